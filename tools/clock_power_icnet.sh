#!/bin/bash
# shader clock / socket power while the ICNet pass (config C4) runs (rocm-smi samples next to a long bench run)
python bench.py --model icnet --steps 3000 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > /dev/null 2>&1 &
BP=$!
sleep 8
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -i "sclk\|mclk\|power\|Max Graphics" | tr "\n" ";" | cut -c1-400; echo
  sleep 0.3
done
wait $BP
echo "--- idle"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr "\n" ";" | cut -c1-300; echo
