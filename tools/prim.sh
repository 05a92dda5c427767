for K in 1024 4096; do
LIBRA_GEMM_KERNEL=256 python tools/gemm_one.py 16384 1024 $K 0 0 20
LIBRA_GEMM_KERNEL=128 python tools/gemm_one.py 2080 1024 $K 0 0 20
LIBRA_GEMM_KERNEL=128 python tools/gemm_one.py 16384 1024 $K 0 0 20
LIBRA_GEMM_KERNEL=128 python tools/gemm_one.py 8192 1024 $K 0 0 20
LIBRA_GEMM_KERNEL=256 python tools/gemm_one.py 18464 1024 $K 0 0 20
LIBRA_GEMM_KERNEL=128 python tools/gemm_one.py 18464 1024 $K 0 0 20
done
