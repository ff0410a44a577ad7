#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P="python3 tools/rsos_probe.py --warm 40 --reps 200 --seconds 600 --channels 8 --only-fused --oracle 0"
ms() { grep -o '"fused_ms": [0-9.]*' | tr '\n' ' '; }
echo "12 waves: mix $($P 2>/dev/null | ms)  plain $($P --plain 2>/dev/null | ms)"
echo "16 waves, two loaders: mix $(SIGOPS_RSOS_NWAVES=16 $P 2>/dev/null | ms)  plain $(SIGOPS_RSOS_NWAVES=16 $P --plain 2>/dev/null | ms)"
echo "helper (17): mix $(SIGOPS_RSOS_NWAVES=17 $P 2>/dev/null | ms)  plain $(SIGOPS_RSOS_NWAVES=17 $P --plain 2>/dev/null | ms)"
echo "helper (17), helper wave absent (wrong results; debug 128 not usable) -- skip"
