#!/usr/bin/env python3
"""Regenerates profiles/INDEX.md: one line per tracked file under profiles/ - what it is, the commit that last touched it and
the claim (DESIGN.md / EXPERIMENTS.md section) it backs.  Descriptions come from the rule table below (first match wins);
a file without a rule is listed as UNDESCRIBED so that it gets one.  usage: python scratch/profiles_index.py"""
import os, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RULES = [
    (r"edge_kernel_traffic\.json", "HBM bytes of the edge kernel by counter, round 1", "DESIGN 4 (roofline table, traffic column)"),
    (r"r05_bench_default_slow_box2\.json", "default bench line of the FINAL library on a box with a slow host and a 16-bit MFMA clock ~10 % down (fp32 47.06, fp16x3 101.7, bf16x3 109.3; training B = 64 rows all ~9.7 ms = that host's enqueue floor); same run as r05_counters.json", "DESIGN 6 (box-to-box spread)"),
    (r"r05_bench_default_slow_box\.json", "default bench line mid-round on a box whose 16-bit MFMA clock was ~6 % lower (fp16 22.2 ns per MFMA)", "DESIGN 6 (box-to-box spread)"),
    (r"r\d+_bench_default.*\.json|r\d+_bench_\d+steps.*\.json|r\d+_bench_(fp32|bf16x3|first_path)\.json|r\d+_bench_3steps\.json",
     "complete JSON line of a `python bench.py` run on a gpurun box", "DESIGN 6 (bench numbers of that round); the `configs` / `next_rows` blocks the driver's tail truncates"),
    (r"r\d+_(fp32|bf16x3|bf16x6|fp16x3|first_path)_T\d+_kernel_stats.*\.csv", "rocprofv3 --kernel-trace --stats of a 50-timestep bench run in that arithmetic", "roofline.achieved / avg_launch_us of the bench line must agree with this average"),
    (r"r01_fp32_firstpath_pmc_.*\.csv", "raw FETCH_SIZE / WRITE_SIZE counter pass of the first round-1 path", "history only"),
    (r"r\d+_counters\.json", "SQ / FETCH_SIZE / WRITE_SIZE counters per kernel family, reduced by scratch/summarize_profiles.py, with the library and source hashes", "bench line `roofline.pmc`, `traffic`; DESIGN 4 tables"),
    (r"r\d+_profile_summary\.txt", "human-readable reduction of the round's counter passes (MFMA-busy, wait fractions, clock, HBM bytes per launch)", "DESIGN 4 / 4b"),
    (r"r\d+_gpu_tests.*\.log|r\d+_smoke.*\.log", "`pytest -m gpu` / smoke output on a gpurun box at the named commit", "parity green at that commit"),
    (r"r\d+_fuzz_parity\.log", "tests/fuzz_parity.py: random forward cases + chains + shard splits against the oracle, every mode of that round", "DESIGN 4 precision modes"),
    (r"r\d+_ablate_.*\.log", "edge kernel with parts switched off (HD_ABLATE bits, debug build), per-launch averages", "DESIGN 4 'where the time goes'; r05: section 12b"),
    (r"r\d+_edge_trace_.*\.log", "per-wave cycle stamps of the edge kernel (HD_ABLATE=16)", "DESIGN 4 / EXPERIMENTS A"),
    (r"r02_f32p_experiment\.log|r02_x6p_experiment\.log", "one-wave-per-SIMD pipelined edge kernels (rejected)", "DESIGN 4b, EXPERIMENTS A"),
    (r"r02_coexec_counters\.log", "SQ_VALU_MFMA_COEXEC_CYCLES per edge kernel", "DESIGN 4b (the fp32 MFMA co-executes with nothing)"),
    (r"r\d+_gamma_spread_.*\.txt", "spread of the learned schedule evaluated in fp32 across hosts / thread counts", "DESIGN 2 'schedule is not reproducible in fp32'"),
    (r"r\d+_small_batch.*\.log", "per-kernel averages of the forward at B = 64 / 16 / 2 in every mode", "DESIGN 4 small-batch paragraphs; r05: 12b"),
    (r"r\d+_r16_sweep.*\.log|r02_gemm16_experiment\.log|r02_direct_sweep\.log", "fp32 node chain at small batches: k_gemm_r16 vs the older chain, ms per forward by batch", "DESIGN 4 k_gemm_r16"),
    (r"r\d+_mix_sweep.*\.log|r02_split_sweep\.log", "k_edge_mixed / k_edge_split break-even sweeps", "DESIGN 4 k_edge_mixed, launch_edge_h rule"),
    (r"r03_edge_valu_variants\.log", "packed-fp32 / LDS-handover variants of the fp32 edge kernel (rejected)", "DESIGN 4"),
    (r"r03_node_phase_trace\.log|r04_node_first_loads_reorder_ab\.log", "phase stamps of k_node_f32 / a reorder A-B (null)", "DESIGN 4 k_node_f32, 12a"),
    (r"r03_power_clock\.log", "rocm-smi clock / power during sustained forwards", "DESIGN 4 (power-limited clock)"),
    (r"r\d+_pmc_tgemm.*\.log|r\d+_pmc_train\.log", "counters of the training GEMM kernels", "DESIGN 10"),
    (r"r05_fuzz_grads_big\.log", "random gradient sweep at widths 128 / 256 on 20-36 molecules: kept pre-activations, fp32 / bf16x6 / fp16x3 per case; 24 cases, no failure", "DESIGN 10 round 5, second half"),
    (r"r05_pmc_train_hbm\.log", "HBM bytes per launch of the training kernels (FETCH_SIZE / WRITE_SIZE passes) against their algorithmic bytes, before / after the XCD-aware placement of the backward stages", "DESIGN 10 round 5, second half"),
    (r"r05_train_xcd_placement\.log", "training step + kernel tables after the XCD-aware placement of the backward stages (fp32 26.5 ms, fp16x3 19.5 ms)", "DESIGN 10 round 5, second half"),
    (r"r05_ab_fused_gamma\.log", "same-box A/B of the B = 16 / 256 training step: schedule network fused vs torch ops (null), loss fused vs torch ops (-1.9 ms at B = 16)", "EXPERIMENTS N"),
    (r"r05_splitk_last_workgroup\.log", "training step with split-K finished by the last workgroup of a tile instead of a reduce launch (2x slower; removed)", "EXPERIMENTS O"),
    (r"r05_ab_small_gemm\.log", "same-box A/B of the training step by batch size: library before / after k_tgemm_small (B = 64: -3 %, B = 128: -3.5 %, B <= 32 host-bound)", "DESIGN 10 round 5, third part"),
    (r"r05_small_gemm_first_run\.log", "training steps B = 16..128 and the host / device split of a B = 16 step with k_tgemm_small (device time 8.76 -> 6.82 ms)", "DESIGN 10 round 5, third part"),
    (r"r05_ab_rebuild_p\.log", "same-box A/B of the fp16x3 training step: P materialised vs rebuilt inside the dW2 kernel (null)", "EXPERIMENTS L"),
    (r"r05_train_b16_kstats_after\.log", "kernel launch counts of the B = 16 training step with the fused loss, fused AdamW and single-slab node dW", "DESIGN 10 round 5, third part"),
    (r"r05_train_b16_after_prep\.log", "B = 16 / 256 training step after hd_edge_prep (733 launches per step at B = 16)", "DESIGN 10 round 5, third part"),
    (r"r05_train_launch_census_b16\.log", "the 701 device launches of one B = 16 training step by kernel and by issuing op, after the round's cuts", "DESIGN 10 round 5, third part; 12"),
    (r"r05_train_host_time_before\.log", "host enqueue time by section and launch count of a B = 16 training step before the fused loss / fused AdamW (12.07 ms, 1,122 launches)", "DESIGN 10 round 5, third part"),
    (r"r05_train_b16_kstats_before\.log", "kernel launch counts of the B = 16 training step before the fused loss", "DESIGN 10 round 5, third part"),
    (r"r05_train_fused_loss_times\.log", "training step by batch size with the fused loss and fused AdamW (B = 16: 8.3-9.7 ms, 867 launches)", "DESIGN 10 round 5, third part"),
    (r"r05_train_keep_ab\.log", "same-box A/B by batch size and arithmetic: pre-activations kept vs recomputed", "DESIGN 10 round 5, second half"),
    (r"r05_train_kstats_final\.log", "kernel tables of the training step on the FINAL library: B = 256 fp16x3, B = 256 fp32, B = 16 fp32 (k_tgemm_small 9-14 us)", "DESIGN 10 round 5"),
    (r"r05_train_kept_pre2\.log", "training step + kernel tables with the kept second-layer pre-activations (fp32 27.0 ms, bf16x6 21.9 ms)", "DESIGN 10 round 5, second half"),
    (r"r05_train_fp16x3\.log", "training step + kernel table with training_precision = fp16x3 (20.4 ms)", "DESIGN 10 round 5, second half"),
    (r"r05_train_kstats\.log", "kernel tables of the training step in both arithmetics after the k_edge_dx rewrite", "DESIGN 10 round 5"),
    (r"r\d+_train.*\.log|r02_loss_host_profile_before\.log|r02_topology_and_loss_host\.log|r03_topology_time\.log|r04_fresh_masks\.log", "training step: per-kernel / per-op times, host-side costs, fresh-mask staging", "DESIGN 10; r05: 10 'round 5'"),
    (r"r04_edge_res_.*", "register-resident persistent fp32 edge kernel k_edge_res (rejected), versions v1-v5 and the A/B", "DESIGN 12a, EXPERIMENTS D"),
    (r"r04_headline_trajectory_modes\.log", "final x / h of a complete T = 1000 run in every mode against the exact-fp32 run", "DESIGN 4 precision modes"),
    (r"r05_stage2_layer_kstats_.*\.log", "stage-2 gcl_full layer, per-kernel averages: before / after the direct path / the rejected agg variant / with the node side on k_node_split_f32", "DESIGN 11 round 5 (0.115 -> 0.086 ms)"),
    (r"r04_stage2_profile\.log|r05_stage2.*", "stage-2 growth step and E_GCL layer, kernel stats", "DESIGN 11"),
    (r"r04_sustained_20steps\.log|r05_sustained.*", "20 timed bench steps per mode (sustained clocks)", "DESIGN 6"),
    (r"r02_concurrent_shards\.log", "two shards on one GPU through two streams", "DESIGN 7"),
    (r"r02_mfma_order\.log", "accumulation order of v_mfma_f32_16x16x4 vs 32x32x2", "DESIGN 4 k_gemm_r16 (bit identity)"),
    (r"r03_small_batch_notes\.log", "notes on the B = 64 decomposition", "DESIGN 4"),
    (r"r05_mb_mfma_stream\.log", "scratch/mb/mstream.hip: ns per fp16 MFMA per SIMD, register operands vs LDS fragments, 1-2 waves per SIMD, whole chip", "DESIGN 12b: the sustained fp16 MFMA rate is 19.7 ns (1.62 GHz under load), not 13.3"),
    (r"r05_mb_wave_specialisation.*\.log", "scratch/mb/ws.hip: matrix-only + vector-only wavefronts on one SIMD (v0: rows without L2 locality / request sunk by hipcc; final: fixed)", "DESIGN 12b: wave specialisation predicts at most -15 %"),
    (r"r05_ablate_fp16x3\.log", "fp16x3 edge kernel with parts ablated, incl. the stream / barrier split (bits 32 / 64)", "DESIGN 12b: the W2 LDS-DMA stream costs 16 %, the barrier 3 %"),
    (r"r05_ab_pgen_spread\.log", "same-box A/B: first-layer operand generation spread over the whole chunk (null)", "EXPERIMENTS H"),
    (r"r05_ab_peel_first_chunk\.log", "same-box A/B of the peeled first chunk (zero C operand)", "DESIGN 12b (-1 %)"),
    (r"r05_copybuffer_probe\.log", "kernel trace with 23 vs 63 forwards: every __amd_rocclr_copyBuffer precedes the first forward kernel", "VERDICT r4 weak 8: the copies are load_numpy_state_dict's, not the forward's"),
    (r"r05_families_.*\.log", "scratch/fwd_families.py: ms per forward and per-family launch averages by batch", "DESIGN 12b / 6"),
    (r"r05_node_split_sweep\.log", "fused k_node<F16> vs the three-launch k_node_split chain by batch size, two runs", "DESIGN 4 k_node_split (threshold 2,048 rows)"),
    (r"r05_f32_node_split_sweep\.log", "exact fp32: headline A/B of k_node_f32 with K quarters, and k_gemm_r16 chain vs k_node_split_f32 chain by batch size (two interleaved repetitions)", "DESIGN 4 k_node_split (fp32 paragraph)"),
    (r"r05_f32_fuse_threshold\.log", "exact fp32: fused k_node_f32 vs the three-launch chain at B = 128 .. 256", "HD_FUSE_MIN_ROWS = 5,400"),
    (r"r05_.*", "round-5 measurement", "DESIGN 0a"),
    (r"r06_digest_cost\.log", "cost of the content digest behind the packed images: bare call and per-call `_forward` with / without it at B = 2 / 16 / 256, both arithmetics", "DESIGN 0a item 6, INTEGRATION 4"),
    (r"r06_fuzz_grads_big\.log", "random gradient sweep at widths 128 / 256 on 20-36 molecules, fp32 / fp16x3 per case, on the round-6 library", "DESIGN 10"),
    (r"r06_gpu_tests_durations\.log", "`pytest -m gpu --durations=40` before the oracle thread cap: one test was 235 s of a 522 s tier on a 128-core host", "tests/conftest.py (OMP_NUM_THREADS)"),
    (r"r06_lib_sha256\.txt", "sha256 of the library every r06 evidence file was collected on", "bench line `roofline.pmc.replayed_from`"),
    (r"r06_train_kstats\.log", "kernel tables of the training step on the round-6 library: B = 256 fp32, B = 256 fp16x3, B = 16 fp32", "DESIGN 10"),
    (r"r06_ablate_fp16x3_no_mfma\.log", "fp16x3 edge kernel of the measurement build: shipped / no MFMA (vector side alone, bit 1024) / matrix side alone / no epilogue, same box", "DESIGN 12 item 2, EXPERIMENTS W: wave specialisation cannot clear 6 %"),
    (r"r06_ab_gemm_rows\.log", "same-box A/B of the node-level training GEMMs: k_tgemm vs the row-resident k_tgemm_rows (slower; removed)", "EXPERIMENTS V"),
    (r"r06_ab_dw2_two_chunks\.log", "same-box A/B of k_dw2_f16 with one vs two chunks in flight (slower; reverted)", "EXPERIMENTS V"),
    (r"r06_.*", "round-6 measurement", "DESIGN 0a"),
]
def commit(path):
    out = subprocess.run(["git", "log", "-1", "--format=%h", "--", path], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    return out or "(uncommitted)"
def describe(name):
    for pat, what, claim in RULES:
        if re.fullmatch(pat, name):
            return what, claim
    return "UNDESCRIBED", "-"
lines = ["# profiles/ - index", "",
         "One line per file: what it is, the commit that last touched it, the claim it backs.  Generated by `scratch/profiles_index.py`.",
         "Files of rounds 1-2 live under `history/` (their numbers are superseded; DESIGN.md / EXPERIMENTS.md cite them as",
         "`profiles/history/r0N_...`).  Raw per-dispatch counter CSVs are not tracked (`.gitignore`); the summaries are.", ""]
for sub in ("", "history"):
    d = os.path.join(ROOT, "profiles", sub)
    names = sorted(n for n in os.listdir(d) if os.path.isfile(os.path.join(d, n)) and n != "INDEX.md")
    lines += [f"## profiles/{sub + '/' if sub else ''}", "", "| file | what | commit | backs |", "|---|---|---|---|"]
    for n in names:
        what, claim = describe(n)
        lines.append(f"| `{n}` | {what} | {commit(os.path.join('profiles', sub, n))} | {claim} |")
    lines.append("")
open(os.path.join(ROOT, "profiles", "INDEX.md"), "w").write("\n".join(lines))
print(sum(1 for l in lines if "UNDESCRIBED" in l), "undescribed of", sum(1 for l in lines if l.startswith("| `")))
