#!/usr/bin/env python3
"""Headline benchmark: samples/s of one full ROCO-MLM training step (BASELINE.json configs[1]):
resnet152 + transformer, num_vis 5, hidden 768, per-GPU batch 16, 224x224, T=32, fp32;
forward + MLM loss + backward + (gradient all-reduce) + Adam, dropout active, train-mode BatchNorm,
inputs resident in HBM.  Prints ONE JSON line (see the driver contract in the task statement).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """threads this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32-input matrix peak
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
B_PER_GPU, T, HW, VOCAB = 16, 32, 224, 30522


CONFIG = 2   # BASELINE.json configs[1] (the metric's config); --config 3 / 4 / 5 = configs[2] / [3] / [4] (tf_efficientnetv2_m
             # + realformer: MLM bs16 | MLM + SupCon on 2N = 32 views | VQA fine-tune with ASLSingleLabel, bs64, T 28)
N_CLASSES = 1552   # vqamed2019 answer classes (config 5)


def make_args():
    from types import SimpleNamespace
    if CONFIG in (3, 4):
        return SimpleNamespace(task="MLM", dataset="roco", transformer_model="realformer",
                               cnn_encoder="tf_efficientnetv2_m", num_vis=5, hidden_size=768, n_layers=4, heads=12,
                               hidden_dropout_prob=0.3, vocab_size=VOCAB, use_relu=False, max_position_embeddings=T,
                               supcon=(CONFIG == 4))
    if CONFIG == 5:
        return SimpleNamespace(task="VQA", dataset="VQA-Med", transformer_model="realformer",
                               cnn_encoder="tf_efficientnetv2_m", num_vis=5, hidden_size=768, n_layers=4, heads=12,
                               hidden_dropout_prob=0.3, vocab_size=N_CLASSES, emb_vocab=VOCAB, use_relu=False,
                               max_position_embeddings=T)
    return SimpleNamespace(task="MLM", dataset="roco", transformer_model="transformer", cnn_encoder="resnet152",
                           num_vis=5, hidden_size=768, n_layers=4, heads=12, hidden_dropout_prob=0.3,
                           vocab_size=VOCAB, use_relu=False, max_position_embeddings=T)


def cpu_baseline(seed):
    """the oracle (CPU restatement, kind 'port') timed on this host at the SAME batch (16): median of 5 training
    steps after 1 warm-up"""
    from oracle import mmbert_oracle as O
    from mmvqa_amd import synth
    torch.manual_seed(seed)
    cores = min(host_cores(), 32)
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} host threads")
    m = O.OracleModel(O.make_args(**vars(make_args()))).train()
    opt = torch.optim.Adam(m.parameters(), lr=2e-5)
    Bc = B_PER_GPU
    img, ids, seg, mask, tgt = synth.roco_batch(Bc, T, HW, VOCAB, seed=seed)

    def step():
        opt.zero_grad()
        loss = O.mlm_loss(m(img, ids, seg, mask), tgt)[0]
        loss.backward()
        opt.step()

    step()
    log("cpu_baseline: warm-up step done")
    times = []
    n = 5
    for _ in range(n):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[n // 2]
    log(f"cpu_baseline: {n} steps, median {med:.2f}s")
    return dict(value=Bc / med, unit="samples/s", cores=cores, kind="port", ms_per_step=med * 1e3,
                sample=f"median of {n} training steps of batch {Bc} (same model, shapes and inputs as the GPU run: "
                       f"fwd + log_softmax/NLL + bwd + Adam, dropout on, train-mode BN) after 1 warm-up")


def per_block(regs, marks, model, a):
    """SURVEY 8(d) per-block figures of ONE profiled step (HIP events per launch, both streams as in the timed steps):
    MFMA-bound blocks as TFLOP/s and fraction of the fp32-MFMA peak, HBM-bound ones as GB/s and fraction of 8 TB/s."""
    heads = 12 if CONFIG == 2 else 8
    L = 4
    M = B_PER_GPU * T

    def gemm(*names):
        ms = sum(regs[n]["igemm"]["ms"] for n in names)
        fl = sum(regs[n]["igemm"]["flops"] for n in names)
        nl = sum(regs[n]["igemm"]["launches"] for n in names)
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return dict(tflops=tf, frac_of_f32_mfma_peak=tf / PEAK_F32_MFMA_TFLOPS, ms=ms, launches=nl, gflop=fl / 1e9)

    def region_ms(n):
        return sum(v["ms"] for v in regs[n].values())

    out = {}
    out["qkv_gemm"] = gemm("qkv")                                  # fused QKV (or RealFormer kqv) GEMM launches: fwd (unless fused below) + dgrad + wgrad
    at = regs["attention"]["attention"]                            # stand-alone attention launches (backward; forward unless fused)
    fu = regs["qkv_attention_fused"]["attention"]                  # qkvattn.hip: projection + attention of a BertLayer, forward, one launch
    if fu["launches"]:
        tf_f = fu["flops"] / (fu["ms"] * 1e-3) / 1e12
        out["qkv_attention_fused_fwd"] = dict(tflops=tf_f, frac_of_f32_mfma_peak=tf_f / PEAK_F32_MFMA_TFLOPS, ms=fu["ms"],
                                              launches=fu["launches"], gflop=fu["flops"] / 1e9,
                                              us_per_launch=fu["ms"] * 1e3 / fu["launches"])
    qk_ms = out["qkv_gemm"]["ms"] + at["ms"] + fu["ms"]
    qk_fl = regs["qkv"]["igemm"]["flops"] + at["flops"] + fu["flops"]
    tf = qk_fl / (qk_ms * 1e-3) / 1e12 if qk_ms > 0 else 0.0
    fwd_ms = fu["ms"] if fu["launches"] else None
    out["qkv_plus_attention_block"] = dict(tflops=tf, frac_of_f32_mfma_peak=tf / PEAK_F32_MFMA_TFLOPS, ms=qk_ms,
                                           gflop=qk_fl / 1e9, target_frac=0.60,
                                           forward=dict(ms=fwd_ms, gflop=fu["flops"] / 1e9,
                                                        frac_of_f32_mfma_peak=(fu["flops"] / (fu["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS))
                                           if fu["launches"] else None,
                                           backward=dict(ms=qk_ms - fu["ms"], gflop=(qk_fl - fu["flops"]) / 1e9,
                                                         frac_of_f32_mfma_peak=((qk_fl - fu["flops"]) / ((qk_ms - fu["ms"]) * 1e-3) / 1e12
                                                                                / PEAK_F32_MFMA_TFLOPS))
                                           if fu["launches"] and qk_ms > fu["ms"] else None,
                                           note="north-star block: QKV projection products + attention, fwd+bwd, 4 layers (forward = one "
                                                "fused launch per layer when `forward` is set)")
    # attention softmax: algorithmic bytes per SURVEY 8(d) = read scores + write probs = 2*B*h*T^2*4 per layer and pass; the
    # stand-alone attention launches are forward + backward (x3) or, with the fused forward, backward only (x2); the
    # tensors are L2-resident at T=32
    passes = 2 if fu["launches"] else 3
    sm_bytes = 2.0 * B_PER_GPU * heads * T * T * 4 * L * passes
    out["attention_softmax"] = dict(ms=at["ms"], launches=at["launches"], algorithmic_mb=sm_bytes / 1e6,
                                    gbps=sm_bytes / (at["ms"] * 1e-3) / 1e9 if at["ms"] > 0 else 0.0,
                                    frac_of_hbm_peak=(sm_bytes / (at["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS) if at["ms"] > 0 else 0.0,
                                    note="stand-alone attention launches only; score/prob tiles live in registers and L2 (T=32): latency-bound, not HBM-bound")
    out["backbone"] = dict(gemm("backbone", "tap"), total_ms_all_kernels=region_ms("backbone") + region_ms("tap") + region_ms("bn_coef"))
    out["taps"] = gemm("tap")
    out["encoder_rest"] = gemm("encoder_rest")
    out["heads_K11"] = dict(gemm("heads"), total_ms_all_kernels=region_ms("heads"))
    if "loss_fwd_a" in marks:
        fwd_ms = marks["loss_fwd_a"].elapsed_time(marks["loss_fwd_b"])
        lb = float(M) * VOCAB * 4
        out["vocab_log_softmax_K12_fwd"] = dict(ms=fwd_ms, algorithmic_mb=lb / 1e6, gbps=lb / (fwd_ms * 1e-3) / 1e9,
                                                frac_of_hbm_peak=lb / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                note="one read of the logits: row max/sum, NLL, argmax (+ the 512-row reduce)")
        k11 = out["heads_K11"]
        tot = k11["total_ms_all_kernels"] + fwd_ms
        tf12 = k11["gflop"] * 1e9 / (tot * 1e-3) / 1e12 if tot > 0 else 0.0
        out["heads_K11_plus_K12"] = dict(ms=tot, tflops=tf12, frac_of_f32_mfma_peak=tf12 / PEAK_F32_MFMA_TFLOPS,
                                         note="head GEMMs fwd+bwd + LN + the loss forward pass (the dlogits pass runs inside backward)")
        adam_ms = marks["adam_a"].elapsed_time(marks["adam_b"])
        ab = 32.0 * model.flat_params.numel()   # read p,g,m,v + write p,m,v + zeroed g
        out["adam"] = dict(ms=adam_ms, algorithmic_gb=ab / 1e9, gbps=ab / (adam_ms * 1e-3) / 1e9,
                           frac_of_hbm_peak=ab / (adam_ms * 1e-3) / 1e9 / PEAK_HBM_GBS)
    bc = regs["bn_coef"]["other"]
    out["bn_coef"] = dict(launches=bc["launches"], ms=bc["ms"])
    out["embed"] = dict(ms=region_ms("embed"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    # knobs of the gradient exchange for the first real multi-GPU sweeps (defaults = what the driver's runs use)
    ap.add_argument("--bucket-mb", type=float, default=64.0, help="size of the all-reduce buckets cut from the flat gradient buffer")
    ap.add_argument("--nccl-algo", type=str, default=None, help="NCCL_ALGO for RCCL (e.g. Ring, Tree)")
    ap.add_argument("--nccl-proto", type=str, default=None, help="NCCL_PROTO (e.g. Simple, LL, LL128)")
    ap.add_argument("--nccl-min-nchannels", type=int, default=None, help="NCCL_MIN_NCHANNELS: more channels use more xGMI links at once")
    ap.add_argument("--overlap-adam", action="store_true",
                    help="A/B switch: Adam per finished gradient range beside the backward pass instead of one launch after it "
                         "(measured time-neutral on configs 2 and 3: DESIGN 7.4)")
    ap.add_argument("--native-comm", action="store_true",
                    help="gradient all-reduce through include/mmvqa_comm.h (mmvqa_allreduce_bucket on an own RCCL communicator and "
                         "stream) instead of torch.distributed.all_reduce")
    a = ap.parse_args()
    for k, v in (("NCCL_ALGO", a.nccl_algo), ("NCCL_PROTO", a.nccl_proto), ("NCCL_MIN_NCHANNELS", a.nccl_min_nchannels)):
        if v is not None:
            os.environ[k] = str(v)         # read by RCCL when the communicator is created (init_process_group below)
    global CONFIG, B_PER_GPU, T
    CONFIG = a.config
    if CONFIG == 4:
        B_PER_GPU = 32     # 2N = 32 views: 16 samples x 2 augmentations per GPU (pretrain/roco_supcon_train.py:137)
    if CONFIG == 5:
        B_PER_GPU, T = 64, 28

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: one process per GPU -- launch with "
                 f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 "
                 f"bench.py --gpus {a.gpus} ...`")
    # rehearsal of the N>1 path on a ONE-GPU box (not a measurement): MMVQA_REHEARSE_GLOO=1 puts every rank on cuda:0
    # and exchanges over gloo; the driver's runs use one GPU per rank over RCCL ("nccl")
    rehearse = bool(os.environ.get("MMVQA_REHEARSE_GLOO"))
    if rehearse:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)

    import mmvqa_amd
    from mmvqa_amd import synth
    from mmvqa_amd.ddp import GradReducer, NativeComm, comm_info, sync_replicas

    torch.manual_seed(1234)            # identical initial weights on every rank
    torch.set_num_threads(min(host_cores(), 16))
    log("building model (random init on the host)")
    model = mmvqa_amd.Model(make_args())
    model.to(dev).train()
    log(f"model on {dev}: {model.flat_params.numel() / 1e6:.1f} M parameters")
    model.set_seed(1234 + rank)
    replica_checksum = sync_replicas(model)    # broadcast from rank 0 + checksum equal on every rank (raises otherwise)
    opt = mmvqa_amd.FusedAdam(model, lr=2e-5)
    native = NativeComm() if (a.native_comm and world > 1 and not rehearse) else None
    red = GradReducer(model.flat_grads, bucket_mb=a.bucket_mb, native=native)
    if world > 1:
        # all-reduce of finished gradient ranges overlaps the backbone backward; `ready` orders RCCL's stream explicitly
        model.set_grad_ready_hook(red.start, with_event=True)
    if a.overlap_adam:
        # Adam of a finished (and, for N > 1, all-reduced) gradient range runs beside the rest of the backward pass
        opt.overlap_backward(red, grad_scale=1.0 / world)
    if CONFIG == 4:
        from mmvqa_amd import train as TR
        va = synth.roco_batch(B_PER_GPU // 2, T, HW, VOCAB, seed=1234 + rank, device=dev)
        vb = synth.roco_batch(B_PER_GPU // 2, T, HW, VOCAB, seed=4321 + rank, device=dev)
        img, ids, seg, mask, tgt = TR.process_tensors((va[0], vb[0]), va[1], vb[1], va[2], va[3], va[4], vb[4])
    elif CONFIG == 5:
        from mmvqa_amd import train as TR
        img, ids, seg, mask, tgt = synth.vqa_batch(B_PER_GPU, T, HW, VOCAB, N_CLASSES, seed=1234 + rank, device=dev)
    else:
        img, ids, seg, mask, tgt = synth.roco_batch(B_PER_GPU, T, HW, VOCAB, seed=1234 + rank, device=dev)

    marks = {}   # torch events around the launches that Python enqueues itself (loss kernels, Adam): same stream

    def step(timed_parts=False):
        ev = (lambda: torch.cuda.Event(enable_timing=True)) if timed_parts else None

        def mark(name):
            if timed_parts:
                e = ev()
                e.record()
                marks[name] = e

        if CONFIG == 4:    # MLM on both views + SupCon over the (all-gathered) view set: mm-vqa_amd/train.py
            loss, _, stats = TR.supcon_step(model, opt, red, world, (img, ids, seg, mask, tgt))
            return [loss.detach()]
        if CONFIG == 5:    # VQA head + ASLSingleLabel
            loss, _ = TR.vqa_step(model, opt, red, world, (img, ids, seg, mask, tgt), mmvqa_amd.asl_loss)
            return [loss.detach()]
        mark("step_a")
        logits = model(img, ids, seg, mask)
        mark("loss_fwd_a")
        loss, _, stats = mmvqa_amd.mlm_loss(logits, tgt)
        mark("loss_fwd_b")
        loss.backward()
        red.allreduce()
        mark("adam_a")
        opt.step(grad_scale=1.0 / world, zero_grad=True)
        mark("adam_b")
        return stats

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    n_tuned = model.tune(img, ids, seg, mask)   # per-shape tile/split-K selection (untimed, like warm-up)
    log(f"tuned {n_tuned} GEMM shapes")
    for i in range(a.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        stats = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = dt / a.steps * 1e3
    log(f"timed {a.steps} steps: {ms:.2f} ms/step")
    value = B_PER_GPU * world * a.steps / dt

    # Roofline of the dominant kernel class: one more step, identical to the timed ones (same two HIP streams), with
    # a HIP event pair around every launch recorded on the stream that launch goes to.  The weight-gradient GEMMs
    # run beside the data-gradient chain, so a launch's duration includes the time it shares the chip: the average
    # agrees with the steady-state rocprofv3 kernel trace (profiles/round2_kernel_stats_steady_cfg2.csv).
    roof = None
    phases = None
    if not a.no_roofline and CONFIG in (2, 3):
        # where the step's wall time goes: one more step as the timed ones run (no per-launch events), torch events on the
        # caller's stream between the phases (every phase ends with that stream joined to the side stream)
        step(timed_parts=True)
        torch.cuda.synchronize()
        ph = lambda a_, b_: marks[a_].elapsed_time(marks[b_])
        phases = dict(forward_ms=ph("step_a", "loss_fwd_a"), loss_ms=ph("loss_fwd_a", "loss_fwd_b"),
                      backward_ms=ph("loss_fwd_b", "adam_a"), adam_ms=ph("adam_a", "adam_b"),
                      note="forward = CNN + taps + encoder + heads; loss = log-softmax / NLL / accuracy pass; backward incl. its "
                           "dlogits pass and the join with the weight-gradient stream")
    if not a.no_roofline:
        model.profile(True)
        step(timed_parts=True)
        torch.cuda.synchronize()
        pr = model.profile_read()
        regs = model.profile_read_regions()
        hbm = model.profile_read_hbm()
        model.profile(False)
        g = pr["igemm"]
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        model.profile(True, serialized=True)   # the same step on ONE stream: every launch has the chip to itself
        step()
        torch.cuda.synchronize()
        gs = model.profile_read()["igemm"]
        regs_alone = model.profile_read_regions()
        model.profile(False)
        ach_alone = gs["flops"] / (gs["ms"] * 1e-3) / 1e12 if gs["ms"] > 0 else 0.0
        traffic, traffic_src = None, None
        for name in ((f"round3_igemm_traffic_cfg{CONFIG}.json",) if CONFIG != 2 else
                     ("round3_igemm_traffic.json", "round2_igemm_traffic.json", "round1_igemm_traffic.json")):
            tf = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tf):   # HBM bytes per launch from the committed rocprofv3 PMC passes
                traffic = json.load(open(tf))["igemm"]["hbm_bytes_per_launch"]
                traffic_src = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/pmc_traffic.py)"
                break
        roof = dict(note="achieved / frac: FLOPs of a launch over that launch's OWN duration, averaged over the step's igemm launches "
                         "as the timed steps run them -- up to three launches share the chip (data-gradient chain, weight gradients, "
                         "tap backward), so a launch's duration includes the time it shares; single_stream = the same step with every "
                         "launch alone on the chip; step_level = the same FLOPs over the wall time of forward + backward",
                    bound="mfma", achieved=ach, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=ach / PEAK_F32_MFMA_TFLOPS, traffic=traffic, traffic_source=traffic_src, kernel="igemm_kernel (fp32 MFMA implicit GEMM)",
                    launches_per_step=g["launches"], avg_launch_us=g["ms"] * 1e3 / max(1, g["launches"]),
                    algorithmic_gflop_per_step=g["flops"] / 1e9,
                    single_stream=dict(achieved=ach_alone, frac=ach_alone / PEAK_F32_MFMA_TFLOPS,
                                       avg_launch_us=gs["ms"] * 1e3 / max(1, gs["launches"]),
                                       note="same step with the second HIP stream disabled: no launch shares the chip"),
                    other_ms_per_step=dict(attention=pr["attention"]["ms"], elementwise=pr["other"]["ms"]),
                    # matrix work outside igemm_kernel (register-resident stem tap, squeeze-excite layers: configs 3-5)
                    matrix_outside_igemm=dict(launches=pr["matrix_other"]["launches"], ms=pr["matrix_other"]["ms"],
                                              gflop=pr["matrix_other"]["flops"] / 1e9))
        if phases is not None:
            roof["phases"] = phases
            wall = phases["forward_ms"] + phases["backward_ms"]
            roof["step_level"] = dict(achieved=g["flops"] / (wall * 1e-3) / 1e12, frac=g["flops"] / (wall * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                      note="GEMM FLOPs of the step over the WALL time of forward + backward (every other kernel, gap and "
                                           "stream join included): unlike `frac` it does not shrink when launches overlap")
        if CONFIG in (2, 3):
            roof["blocks"] = per_block(regs, marks, model, a)
            # the same table from the one-stream pass: every launch alone on the chip.  In the step the weight gradients run
            # beside the data gradients (encoder included since round 3), which stretches each launch's own duration and
            # with it every figure of `blocks`, although the step gets shorter
            roof["blocks_single_stream"] = per_block(regs_alone, {}, model, a)
        # the HBM-bound kernels one by one: algorithmic bytes (every tensor read / written once, fp32) over the HIP-event
        # time of their launches in this step, against the 8 TB/s roofline
        roof["hbm_kernels"] = {k: dict(launches=v["launches"], ms=v["ms"], algorithmic_gb=v["bytes"] / 1e9,
                                       gbps=v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0,
                                       frac_of_hbm_peak=(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS) if v["ms"] > 0 else 0.0)
                               for k, v in hbm.items()}

    names = {2: ("ROCO-MLM pretrain (resnet152+transformer, bs16/GPU, 224^2, seq32)",
                 "pretrain/roco_train.py MLM-only: resnet152 + transformer(4 layers, 12 heads), num_vis 5, hidden 768, vocab "
                 "30522, per-GPU batch 16, 224x224, T=32; fwd + log_softmax/NLL + bwd + grad all-reduce + Adam"),
             3: ("ROCO-MLM pretrain (tf_efficientnetv2_m+realformer, bs16/GPU, 224^2, seq32)",
                 "pretrain/roco_train.py MLM-only: tf_efficientnetv2_m + realformer(4 layers, 8 heads), num_vis 5, hidden 768, "
                 "vocab 30522, per-GPU batch 16, 224x224, T=32; fwd + log_softmax/NLL + bwd + grad all-reduce + Adam"),
             4: ("ROCO MLM+SupCon pretrain (tf_efficientnetv2_m+realformer, 2N=32 views/GPU, 224^2, seq32)",
                 "pretrain/roco_supcon_train.py --con_task=supcon: tf_efficientnetv2_m + realformer + SupCon head, 16 samples x 2 "
                 "views per GPU, 224x224, T=32; fwd + MLM loss + all-gather of the views + SupCon loss + bwd + grad "
                 "all-reduce + Adam"),
             5: ("VQA-Med-2019 fine-tune (tf_efficientnetv2_m+realformer, ASLSingleLabel, bs64/GPU, 224^2, seq28)",
                 "vqamed2019/train.py --loss=ASLSingleLabel: tf_efficientnetv2_m + realformer, VQA head with 1552 classes, "
                 "per-GPU batch 64, 224x224, T=28; fwd + ASL + bwd + grad all-reduce + Adam")}
    metric, wl = names[CONFIG]
    last = stats[0]
    out = dict(metric="samples/sec " + metric, value=value, unit="samples/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
               ms_per_step=ms, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload=wl + "; dropout on, train-mode BN; random-init weights",
                           global_batch=B_PER_GPU * world, seq_len=T, parallelism=f"dp{world}",
                           samples_per_s_per_gpu=value / world, final_loss=float(last.detach() if hasattr(last, "detach") else last),
                           comm=dict(comm_info(red), replica_checksum=replica_checksum, native_comm=native is not None,
                                     rehearsal_on_one_gpu=rehearse)))
    if roof is not None:
        out["roofline"] = roof
    if rank == 0 and world == 1 and not a.no_cpu_baseline and CONFIG == 2:
        out["cpu_baseline"] = cpu_baseline(1234)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
