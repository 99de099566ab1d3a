cd $GRAFT_REPO_ROOT
timeout 700 python tools/dev/soak_timed.py 6900 7200 15 > gpurun_out/soak_final_timed.log 2>&1; tail -2 gpurun_out/soak_final_timed.log
timeout 900 python tools/dev/soak_zones.py 6000 8000 local > gpurun_out/soak_final_zones.log 2>&1; tail -2 gpurun_out/soak_final_zones.log
timeout 500 python tools/dev/soak_quad.py 600 660 > gpurun_out/soak_final_quad.log 2>&1; tail -2 gpurun_out/soak_final_quad.log
timeout 500 python tools/dev/soak_tail2.py 1300 1360 > gpurun_out/soak_final_tail2.log 2>&1; tail -2 gpurun_out/soak_final_tail2.log
