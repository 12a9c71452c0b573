for cl in 0 1; do
  echo "== CRH_CLAMP_GRID=$cl"
  for cfg in C2 C3; do CRH_CLAMP_GRID=$cl python bench.py --config $cfg --no-cpu --no-interactive --steps 3 2>/dev/null | tail -1 | cut -c1-110; done
  CRH_CLAMP_GRID=$cl python tools/bench_two_level.py 10 32 2>/dev/null | tail -1
done
