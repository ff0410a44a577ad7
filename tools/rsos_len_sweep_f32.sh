# Float32 signal all the way, 8 channels at 44.1 -> 48 kHz + band-pass: one launch (k_rsos) against K3 + K2, and what the planner picks
for secs in 20 30 46 60 75 120; do
  fr=$(python3 -c "print(int(44100*$secs))")
  a=$(SIGOPS_RSOS_MINGROUPS=1 F32=1 FRAMES=$fr ONLY="ToFramerate 48k | Filt" timeout 60 python3 tools/operator_matrix.py 2>/dev/null | awk '{print $5, $9}')
  b=$(SIGOPS_NO_RSOS=1 F32=1 FRAMES=$fr ONLY="ToFramerate 48k | Filt" timeout 60 python3 tools/operator_matrix.py 2>/dev/null | awk '{print $5, $9}')
  c=$(F32=1 FRAMES=$fr ONLY="ToFramerate 48k | Filt" timeout 60 python3 tools/operator_matrix.py 2>/dev/null | awk '{print $5, $9}')
  echo "$secs s: fused $a | two kernels $b | planner's choice $c"
done
