# timing-only ablation of the config-5 step's GEMM kernels (GPU box): which phase bounds the two 21-GFLOP launches
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "" "-DMMS_ABLATE_MMA" "-DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE" "-DMMS_ABLATE_GLOAD"; do
  (cd $R && MMS_CXXFLAGS="$v" python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
  echo "variant [$v]"; STEPS=30 bash $R/tools/prof_c5.sh 2>/dev/null | grep -E "g32x16x1|g16x79x1|g8x16x4|g32x8x1"
done
