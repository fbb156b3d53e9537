#!/bin/bash
# sample sclk / power while the sampler runs: is the edge kernel power-limited?
( timeout 300 python bench.py --steps 6 --warmup 1 > /tmp/bench.log 2>&1 ) &
BP=$!
for i in $(seq 1 400); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  L=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -e 's/.*(\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*(W): \(.*\)/power \1/' | tr '\n' ' ')
  echo "$i $L"
  sleep 0.3
done | awk '{k=$3" "$5; c[k]++} END {for (k in c) print c[k], "samples: sclk", k, "W"}' | sort -rn | head -20
wait $BP
tail -c 300 /tmp/bench.log
