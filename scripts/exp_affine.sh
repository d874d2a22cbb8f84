cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pin1024.py tests/test_gpu_table.py tests/test_gpu_engine.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
STEPS=8 bash scripts/exp_build.sh affine_x 1024
STEPS=8 bash scripts/exp_build.sh affine_x_512 512
