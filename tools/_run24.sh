cd $GRAFT_REPO_ROOT
for ks in "1,8" "1,12" "1,16" "1,24"; do
  echo "== IADR1_DECODE_KS=$ks: $(IADR1_DECODE_KS=$ks timeout 600 python tools/decode_mask_probe.py 192 2>&1 | grep 'decode on' | tail -1)"
done
