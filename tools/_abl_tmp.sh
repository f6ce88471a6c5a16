cd /root/repo
timeout -k 10 300 python -m pytest tests/test_ann_fused_gpu.py -x -q 2>&1 | tail -3
for m in 0 128; do
  BG_LIB_PATH=/root/repo/1d-burgers-equation-roms_amd/build/libabl_$m.so python tools/time_ann_fused.py --batch 512 --steps 20 2>&1 | tail -1
done
python tools/time_ann_fused.py --batch 2048 --steps 40 | tail -1
