# per-rank share time vs number of wave queues (FIREWORK_WAVES): multiples of 28672 = lcm(7, 4) waves/SIMD x 1024 SIMDs fill whole rounds of both k_extend_linear and k_shade
for W in ${SWEEP:-14336 28672 43008 57344 65536 86016 114688 131072}; do
  echo "FIREWORK_WAVES=$W"; SHARE_WORLDS=${SHARE_WORLDS:-4,8} FIREWORK_WAVES=$W python tools/share.py | cut -c1-64,128-175
done
