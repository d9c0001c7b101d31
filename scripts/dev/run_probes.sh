# Builds and runs the micro-benchmarks the DESIGN.md claims rest on; outputs go to gpurun_out/probes/ (copied to profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/probes; mkdir -p $O /tmp/pb
for p in probe_atomics3 probe_atomics4 probe_scatter_stores probe_atomics2 probe_gather_pairs; do
  hipcc --offload-arch=gfx950 -O3 -o /tmp/pb/$p $R/scripts/dev/$p.hip 2> /tmp/pb/$p.build.log && timeout 120 /tmp/pb/$p > $O/$p.txt 2>&1
  echo "== $p rc=$?"; tail -3 $O/$p.txt
done
timeout 300 python3 $R/scripts/dev/probe_encode_bwd_levels.py > $O/probe_encode_bwd_levels.txt 2>&1; echo "== levels rc=$?"
timeout 300 python3 $R/scripts/dev/probe_encode_bwd_binned.py > $O/probe_encode_bwd_binned.txt 2>&1; echo "== binned rc=$?"; cat $O/probe_encode_bwd_binned.txt
