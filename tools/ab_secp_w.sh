for rep in 1 2; do for w in 243 253; do
  NCG_SECP_W=$w NCG_LIB=$PWD/tools/_build/libncg_ab.so timeout 300 python bench.py --workload secp256k1 --no-cpu-baseline --no-live-pmc --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('secp W', $w, round(d['ms_per_step'],3))"
done; done
