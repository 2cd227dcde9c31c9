# product upfirdn2d kernel through the C ABI: strip height x LDS hand-off, headline call and the in-step 96-sample FIRs
cd $GRAFT_REPO_ROOT
for cfg in "32 257 1" "96 256 2" "96 257 1"; do
  for st in 8 16 32; do for lds in 0 1; do
    echo "== $cfg strip $st lds $lds"
    SGV_LANES_STRIP=$st SGV_LANES_LDS=$lds timeout 60 tools/ufd_lab $cfg 2>&1 | grep -E "libsgv|V3 cols dwordx4 PF4 NT1 strip 8|copy2 ntl1"
  done; done
done
