R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for v in "-DMMS_ABLATE_MMA -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE" "-DMMS_ABLATE_MMA -DMMS_ABLATE_GLOAD -DMMS_ABLATE_SSTORE -DMMS_ABLATE_STATS"; do
  (cd $R && MMS_CXXFLAGS="$v" python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
  rm -rf /tmp/abl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -- python3 $R/tools/prof_conv3.py 0 30 > /dev/null 2>&1
  f=$(ls /tmp/abl/*/*_kernel_stats.csv | head -1)
  echo "variant [$v]: $(grep 'Conv3FwdOp' $f | awk -F, '{printf "avg %.1f us min %.1f", $4/1000, $6/1000}')"
done
(cd $R && python -m multimodal_survival_prediction_amd._build --force > /dev/null 2>&1)
