# what power / clock telemetry is readable on the GPU box (round 6)
echo "== rocm-smi"; timeout 20 rocm-smi --showpower --showclocks 2>&1 | head -30
echo "== amd-smi"; timeout 20 amd-smi metric --power --clock 2>&1 | head -40
echo "== python amdsmi"; python -c "import amdsmi; print('amdsmi ok', amdsmi.__file__)" 2>&1 | tail -1
echo "== sysfs"; for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_input /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input /sys/class/drm/card*/device/pp_dpm_sclk; do [ -e $f ] && { echo $f; cat $f | head -12; }; done
