cd $GRAFT_REPO_ROOT
bash scripts/exp_build.sh base 1024
bash scripts/exp_build.sh zx1_x_on_aux 1024 -DSDC_EXP_ZX=1
bash scripts/exp_build.sh zx2_dummy_z_beside_x 1024 -DSDC_EXP_ZX=2
