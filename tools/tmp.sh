python -m pytest tests/test_gpu_parity.py -x -q -k "inverse or dilute_ffield" 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t3 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
grep -h "inv_\|a_kspace\|a_real" gpurun_out/t3/*/*kernel_stats.csv | sed 's/"void conp::\|"conp:://; s/(.*)"//' | cut -c1-120
