R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" "-DC3W_NO_LOAD" "-DC3W_NO_MFMA" "-DC3W_NO_READ" "-DC3W_NO_LOAD -DC3W_NO_READ" "-DC3W_NO_LOAD -DC3W_NO_READ -DC3W_NO_BARRIER" "-DC3W_NO_LOAD -DC3W_NO_MFMA"; do
  (cd $R && MMS_CXXFLAGS="$v" python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
  echo "variant [$v]: b0x10 $(MMS_CONV3W_MT=2 python3 $R/tools/prof_conv3bwdw.py 0 40 10 1024 2>/dev/null | head -1) | b0x5/512 $(MMS_CONV3W_MT=2 python3 $R/tools/prof_conv3bwdw.py 0 40 5 512 2>/dev/null | head -1)"
done
