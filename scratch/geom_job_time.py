"""sample_batches(256, 4) with GEOM-histogram sizes: merged device batches (merge_edges) vs the reference's loop; fp32."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
m = bench.build_model(256, 6, 1000, dev, 0, 1)
m.dynamics.precision = prec
for merge, edges in ((2048, 225_000), (2048, 450_000), (2048, 900_000), (0, 0)):
    m.merge_batches, m.merge_edges = merge, edges
    for rep in range(2):
        torch.manual_seed(2022)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res, _ = m.sample_batches(256, 8, dev)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{prec} merge_batches={merge} merge_edges={edges}: {dt:.3f} s per job of 2048 molecules = {2048/dt:.1f} molecules/s (mean n {sum(r['x'].shape[0] for r in res)/2048:.2f})")
