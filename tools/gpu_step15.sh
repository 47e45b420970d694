set -x
O=gpurun_out/s15; rm -rf $O; mkdir -p $O
for d in 0 1 2 3 4 8 12 16 32 48 63; do
  EV_ATTN_DBG=$d EV_SP_NOVOC=1 python tools/shape_profile.py 64 $O/shape_$d.txt > $O/shape_$d.log 2>&1
  echo "dbg=$d $(grep 'lnqkv  128' $O/shape_$d.txt | tr '\n' '|')"
done
