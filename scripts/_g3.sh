mkdir -p gpurun_out/r02c
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or x3_image" > gpurun_out/r02c/gemm_tests.log 2>&1; tail -15 gpurun_out/r02c/gemm_tests.log
timeout 300 python scripts/gemm_table.py bf16x3 > gpurun_out/r02c/gemm_table_x3.txt 2>&1; grep -v amdgpu.ids gpurun_out/r02c/gemm_table_x3.txt | tail -40
