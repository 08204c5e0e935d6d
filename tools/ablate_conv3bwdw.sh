R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" "-DMMS_ABLATE_FLUSH" "-DMMS_ABLATE_FLUSH -DMMS_ABLATE_MMA" "-DMMS_ABLATE_FLUSH -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE"; do
  (cd $R && MMS_CXXFLAGS="$v" python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
  echo "variant [$v]: b0 $(python3 $R/tools/prof_conv3bwdw.py 0 40 10 2>/dev/null | head -1) | b1 $(python3 $R/tools/prof_conv3bwdw.py 1 100 10 2>/dev/null | head -1) | b2 $(python3 $R/tools/prof_conv3bwdw.py 2 200 10 2>/dev/null | head -1)"
done
(cd $R && python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
