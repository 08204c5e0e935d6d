R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" "-DC0_NO_EPI" "-DC0_NO_MMA" "-DC0_NO_STAGE" "-DC0_NO_MMA -DC0_NO_EPI" "-DC0_NO_STAGE -DC0_NO_EPI"; do
  (cd $R && MMS_CXXFLAGS="$v" python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
  echo "variant [$v]: $(python3 $R/tools/prof_conv0bw.py 5 20 2>/dev/null) | $(python3 $R/tools/prof_conv0bw.py 1 20 2>/dev/null)"
done
(cd $R && python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
