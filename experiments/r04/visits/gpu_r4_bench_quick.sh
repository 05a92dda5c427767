cd /root/repo 2>/dev/null || true
timeout 150 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -4 | cut -c1-400
