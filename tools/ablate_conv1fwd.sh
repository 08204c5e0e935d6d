R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" "-DMMS_ABLATE_STATS" "-DMMS_ABLATE_MMA" "-DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE" "-DMMS_ABLATE_MMA -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE" "-DMMS_ABLATE_MMA -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE -DMMS_ABLATE_STATS" "-DMMS_ABLATE_MMA -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE -DMMS_ABLATE_STATS -DMMS_ABLATE_SETUP"; do
  (cd $R && MMS_CXXFLAGS="$v" python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
  echo "variant [$v]: $(python3 $R/tools/prof_conv1fwd.py 2 640 5 2>/dev/null) | $(python3 $R/tools/prof_conv1fwd.py 3 768 5 2>/dev/null) | $(python3 $R/tools/prof_conv1fwd.py 1 320 5 2>/dev/null)"
done
(cd $R && python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
