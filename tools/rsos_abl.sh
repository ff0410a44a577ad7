# ablations / geometry sweeps of the fused resampler -> IIR kernel on the headline pipeline (600 s x 8 ch)
for d in ${RSOS_DEBUGS:-0 1 2 3}; do echo "debug=$d"; SIGOPS_RSOS_DEBUG=$d timeout 300 python3 tools/rsos_probe.py --seconds 600 --only-fused --oracle 0 --reps 10 2>&1 | grep -o '"fused_ms": [0-9.]*'; done
