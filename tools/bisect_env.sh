for v in NONE SIGOPS_NO_WINDOW_ALIAS SIGOPS_NO_WARM_START SIGOPS_SINGLE_STREAM SIGOPS_K1_NOCHAIN SIGOPS_NO_GRAPH; do
  echo "== $v"; env $v=1 python tools/soak_repro.py a 1396 7 2>&1 | grep "relerr\|engine\|differing" | cut -c1-200
done
