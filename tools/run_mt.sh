# A/B of the two conv3 weight-gradient kernels (GPU box): block, reps, models per launch, rows per chunk
for cfg in "0 40 10 1024" "0 40 5 512" "0 40 8 1024" "0 40 4 512"; do
  for mt in 0 2; do echo "cfg [$cfg] MT=$mt: $(MMS_CONV3W_MT=$mt python3 tools/prof_conv3bwdw.py $cfg 2>/dev/null | head -1)"; done
done
