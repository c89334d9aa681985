timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "strided or pack or x3_image or gemm_big_x3" 2>&1 | tail -3
bash scripts/_prof.sh 40 | grep -E "steps=|strided|x3_image"
