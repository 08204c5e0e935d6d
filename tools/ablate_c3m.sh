R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "-DC3M_TAPS=26" "-DC3M_TAPS=18" "-DC3M_TAPS=8" "-DC3M_TAPS=0"; do
  (cd $R && MMS_CXXFLAGS="$v" python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
  echo "variant [$v]: $(python3 $R/tools/prof_conv3fwd_group.py 0 10 2>/dev/null)"
done
(cd $R && python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
