cd /root/repo
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02_all_gpu.log 2>&1; echo rc=$? >> gpurun_out/r02_all_gpu.log
tail -4 gpurun_out/r02_all_gpu.log
