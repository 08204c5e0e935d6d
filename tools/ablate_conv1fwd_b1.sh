# timing-only ablation of conv1 forward at the block-1 / block-2 shapes of a 10-model group (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" "-DMMS_ABLATE_STATS" "-DMMS_ABLATE_SETUP" "-DMMS_ABLATE_MMA" "-DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE" "-DMMS_ABLATE_MMA -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE" "-DMMS_ABLATE_MMA -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE -DMMS_ABLATE_STATS -DMMS_ABLATE_SETUP"; do
  (cd $R && MMS_CXXFLAGS="$v" python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
  echo "variant [$v]: $(python3 $R/tools/prof_conv1fwd.py 0 64 10 2>/dev/null) | $(python3 $R/tools/prof_conv1fwd.py 0 224 10 2>/dev/null) | $(python3 $R/tools/prof_conv1fwd.py 1 320 10 2>/dev/null)"
done
