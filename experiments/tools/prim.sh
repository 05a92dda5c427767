for K in 1024 4096; do
GEMM_TILE=2 python tools/gemm_one.py 16384 1024 $K 0 0 20
GEMM_TILE=1 python tools/gemm_one.py 2080 1024 $K 0 0 20
GEMM_TILE=1 python tools/gemm_one.py 16384 1024 $K 0 0 20
GEMM_TILE=1 python tools/gemm_one.py 8192 1024 $K 0 0 20
GEMM_TILE=2 python tools/gemm_one.py 18464 1024 $K 0 0 20
GEMM_TILE=1 python tools/gemm_one.py 18464 1024 $K 0 0 20
done
