cd $GRAFT_REPO_ROOT
bash scripts/exp_build.sh base 1024 -DSDC_SPECZ_KY2=0
bash scripts/exp_build.sh ky2_w4_hoist 1024
bash scripts/exp_build.sh ky2_w4_nohoist 1024 -DSDC_KY2_HOIST=0
bash scripts/exp_build.sh ky2_w3_nohoist 1024 -DSDC_KY2_HOIST=0 -DSDC_KY2_WAVES=3
bash scripts/exp_build.sh ky2_w3_hoist 1024 -DSDC_KY2_WAVES=3
