# development aid: the second battery of soaks on the final tree (profiles/soak_r06.txt, last block)
cd $GRAFT_REPO_ROOT
timeout 900 python tools/dev/soak_mid.py 750 800 > gpurun_out/soak2_mid.log 2>&1; tail -1 gpurun_out/soak2_mid.log
SOAK_MISSING=1 timeout 600 python tools/dev/soak_timed.py 7200 7400 15 > gpurun_out/soak2_nan.log 2>&1; tail -1 gpurun_out/soak2_nan.log
timeout 600 python tools/dev/soak_zones.py 200 400 > gpurun_out/soak2_zones_plans.log 2>&1; tail -1 gpurun_out/soak2_zones_plans.log
timeout 900 python tools/dev/soak_zones.py 8000 12000 local > gpurun_out/soak2_zones_local.log 2>&1; tail -1 gpurun_out/soak2_zones_local.log
timeout 600 python tools/dev/soak_quad.py 660 760 > gpurun_out/soak2_quad.log 2>&1; tail -1 gpurun_out/soak2_quad.log
timeout 600 python tools/dev/soak_tail2.py 1360 1460 > gpurun_out/soak2_tail2.log 2>&1; tail -1 gpurun_out/soak2_tail2.log
