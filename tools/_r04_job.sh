for rep in 1 2; do
for spec in "retire:" "noretire:poll_retire=0" "nopoll:poll_long=0"; do
  label="${spec%%:*}"; cfg="${spec#*:}"; flags=""; [ -n "$cfg" ] && flags="--configure $cfg"
  for mode in "" "--host-positions"; do
    v=$(timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-baseline none --no-extra-passes $flags $mode 2>/dev/null | python3 -c "import json,sys; print('%.1f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "rep$rep $label ${mode:-resident}: $v evals/s"
  done
done
done
