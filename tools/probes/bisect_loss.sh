ALL="SALSA_HIP_CONV_1X1 SALSA_HIP_STEM_WRW SALSA_FUSED_SKIP SALSA_FILTER_BANK SALSA_HIP_BN_POOL"
for on in $ALL; do
  envs=""
  for v in $ALL; do if [ $v != $on ]; then envs="$envs $v=0"; fi; done
  echo "only $on:"; env $envs python tools/probes/loss_curve.py 2>/dev/null | tail -1
done
