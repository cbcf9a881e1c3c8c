for rh in 12 16 20; do for tb in 1 2 3; do for tr in 6 8; do
echo -n "bwd rh=$rh b2=$tb rh2=$tr: "
SMD_BWD_RH=$rh SMD_BWD_TAPER_B=$tb SMD_BWD_TAPER_RH=$tr python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_kernel_ms'], d['roofline_bwd']['avg_kernel_ms'])"
done; done; done
