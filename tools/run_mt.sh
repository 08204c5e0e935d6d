# A/B of the two conv3 weight-gradient kernels (GPU box): block, reps, models per launch, rows per chunk, form (MmsDnOpts.conv3w_mt)
for cfg in "0 40 10 1024" "0 40 5 512" "0 40 8 1024" "0 40 4 512"; do
  for mt in -1 2; do echo "cfg [$cfg] conv3w_mt=$mt: $(python3 tools/prof_conv3bwdw.py $cfg $mt 2>/dev/null | head -1)"; done
done
