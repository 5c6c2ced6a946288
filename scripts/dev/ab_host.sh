for v in "" "DUN_AB_COPY=old" "DUN_AB_D2H=old" "DUN_AB_COPY=old DUN_AB_D2H=old" ""; do
  echo "== $v"; env $v timeout -k 10 100 python bench.py --workload c2_uci --steps 3000 --warmup 50 --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
