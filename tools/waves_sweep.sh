for W in 8192 16384 32768 65536 131072 262144; do
  echo "FIREWORK_WAVES=$W split"; FIREWORK_SPLIT=1 FIREWORK_WAVES=$W python tools/share.py | cut -c1-60,100-
done
