timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
BG_B=64 BG_ONLY="encout,ffn2" BG_ROUNDS=3 timeout 120 python tools/bench_gemm.py 2>&1 | grep -v "^total\|amdgpu.ids"
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | grep metric | cut -c60-110
