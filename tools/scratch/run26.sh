cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp QV_GRAPH_CACHE=/tmp/qv_graph_1m.npz
python3 tools/dev_hnsw_r06.py 64 128 1 > /dev/null 2>&1
ls /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -3
for i in 1 2 3 4 5 6; do
  ( while true; do for c in /sys/class/drm/card*/device; do
      s=$(grep '\*' $c/pp_dpm_sclk 2>/dev/null | tr '\n' ' '); m=$(grep '\*' $c/pp_dpm_mclk 2>/dev/null | tr '\n' ' '); f=$(grep '\*' $c/pp_dpm_fclk 2>/dev/null | tr '\n' ' ')
      p=$(cat $c/hwmon/hwmon*/power1_average 2>/dev/null || cat $c/hwmon/hwmon*/power1_input 2>/dev/null); echo "sclk[$s] mclk[$m] fclk[$f] power_uW[$p]"; done; sleep 0.25; done > /tmp/clk_$i.log 2>&1 ) &
  SP=$!
  python3 tools/dev_hnsw_r06.py 32768 128 25 2>&1 | grep nq | cut -c1-40
  kill $SP
  echo "  samples: $(wc -l < /tmp/clk_$i.log)"; sort /tmp/clk_$i.log | uniq -c | sort -rn | head -6
done
