#!/bin/bash
# Which clock / power read-outs an ordinary user has on the GPU box (bench.py's in-kernel clock needs none of them;
# this records what sysfs / amd-smi show beside it).   bash tools/probe_clock_sources.sh > gpurun_out/clock_sources.txt
for f in /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input \
         /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_input \
         /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap; do
  if [ -r "$f" ]; then echo "== $f"; head -12 "$f"; else echo "-- not readable: $f"; fi
done
echo "== rocm-smi --showclocks --showpower"; timeout 30 rocm-smi --showclocks --showpower 2>&1 | head -40
echo "== amd-smi metric -c -p"; timeout 30 amd-smi metric -c -p 2>&1 | head -60
