echo "== sysfs"; for c in /sys/class/drm/card*/device; do echo $c; ls $c/hwmon/ 2>/dev/null; for h in $c/hwmon/hwmon*; do ls $h | tr '\n' ' '; echo; for f in power1_average power1_input freq1_input freq1_label power1_cap; do [ -r $h/$f ] && echo "$f=$(cat $h/$f)"; done; done; cat $c/numa_node 2>/dev/null; cat $c/local_cpulist 2>/dev/null; cat $c/pp_dpm_sclk 2>/dev/null | head -5; done
echo "== which"; which amd-smi rocm-smi; nproc; id
echo "== amd-smi timing"; ( time amd-smi metric -g 0 --power --clock --json ) 2>&1 | tail -60
echo "== rocm-smi timing"; ( time rocm-smi --showpower --showclocks --json ) 2>&1 | tail -20
echo "== lscpu"; lscpu | grep -i "numa\|model name\|^CPU(s)"
