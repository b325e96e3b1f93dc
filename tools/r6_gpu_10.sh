#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_10; mkdir -p $O
{ for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap_max /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap_default /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/power_dpm_force_performance_level; do echo "== $f"; cat $f 2>&1 | head -12; done
  rocm-smi --showpower --showmaxpower --showperflevel --showclocks 2>&1 | head -40
  rocm-smi --showtemp 2>&1 | head -20; } > $O/power_caps.txt 2>&1
cat $O/power_caps.txt | head -70
