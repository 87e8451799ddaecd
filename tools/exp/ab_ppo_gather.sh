# A/B on one box: PPO + AMP optimiser step (bench's policy.ppo leg), gradients gathered into the flat bucket by one launch (round 6) vs
# autograd's accumulation launch per parameter; and the two-piece backward GEMMs (EMLOCO_BWD_PIECES)
for rep in 1 2; do for g in 0 1; do for p in 3 2; do
  EMLOCO_PPO_GATHER_GRADS=$g EMLOCO_BWD_PIECES=$p python tools/exp/ppo_epoch.py 2 2>&1 | grep update_ms | sed "s/^/gather=$g bwd_pieces=$p: /" | cut -c1-400
done; done; done
