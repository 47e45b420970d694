#!/usr/bin/env python3
"""EmojiVoice hot-path benchmark (BASELINE.json: audio-seconds per second per GPU and real-time
factor, 10 Euler steps, 22.05 kHz, batch 64 synthetic 6-s utterances).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = CFM decode (n Euler steps of the U-Net estimator) + HiFi-GAN on one batch of B utterances per
GPU (inputs resident in HBM, z given, weights resident), plus — for N > 1 — the single RCCL all-gather that
collates the waveforms.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

SR, HOP = 22050, 256
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0
# SURVEY.md §8(d), measured on the reference modules: FLOPs and layer-granular activation bytes per 6-s utterance
ALG_FLOP_PER_AUDIO_S = 62.5e9
ALG_BYTES_PER_AUDIO_S = 520e6


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("EV_CPU_THREADS", "32"))))


def make_inputs(B_global, T, lo, hi, device):
    """Config 2/3 inputs: global tensors from fixed seeds; this rank keeps rows [lo, hi)."""
    from emojivoice_amd.emoji import EMOJI_MAPPING

    mu = torch.randn(B_global, 80, T, generator=torch.Generator().manual_seed(1234))[lo:hi]
    z = torch.randn(B_global, 80, T, generator=torch.Generator().manual_seed(1235))[lo:hi]
    ids = torch.tensor(sorted(EMOJI_MAPPING.values()))
    pick = torch.randint(0, len(ids), (B_global,), generator=torch.Generator().manual_seed(1236))[lo:hi]
    spk_ids = ids[pick]
    lengths = torch.full((hi - lo,), T, dtype=torch.int64)
    return mu.to(device), z.to(device), spk_ids.to(device), lengths.to(device)


def cpu_baseline(sd, voc_sd, mu, z0, spk, n_ode):
    """The CPU oracle (restatement of the reference, pinned by reference-generated goldens) timed on this
    box's host cores on a bounded sample (the first rows) of the same workload.  ``z0`` is already scaled
    by the temperature."""
    from emojivoice_amd import weights as W
    from oracle import matcha_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    sample_b, _, T = mu.shape
    log(f"[bench] cpu baseline: oracle on {cores} host threads, B={sample_b} ...")
    mask = torch.ones(sample_b, 1, T)

    def run(n):
        with torch.inference_mode():
            dec = O.solve_euler(sd, z0[:n], mu[:n], mask[:n], n_ode, spk[:n] if spk is not None else None)
            mel = O.denormalize(dec, sd["mel_mean"], sd["mel_std"])
            return mel, O.hifigan_forward(voc_sd, mel, W.HIFIGAN_V1)

    run(1)                                   # warm-up (thread pool, oneDNN primitives): one utterance, ~1-2 s
    times = []
    for _ in range(2):                       # two timed calls on the full sample, the faster one is reported
        t0 = time.perf_counter()
        mel, wav = run(sample_b)
        times.append(time.perf_counter() - t0)
    dt = min(times)
    audio_s = sample_b * T * HOP / SR
    return {"value": round(audio_s / dt, 3), "unit": "audio_s/s", "cores": cores, "kind": "port",
            "sample": f"first {sample_b} utterances of the batch (T={T} frames), {n_ode} Euler steps + HiFi-GAN; 1-utterance warm-up, "
                      f"best of 2 calls ({times[0]:.1f} / {times[1]:.1f} s wall)"}, mel, wav


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=516, help="mel frames per utterance (516 = 5.99 s)")
    ap.add_argument("--ode-steps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="run CFM and HiFi-GAN of each batch back to back on one stream")
    ap.add_argument("--cpu-sample", type=int, default=8)
    args = ap.parse_args()

    from emojivoice_amd import dist as D
    from emojivoice_amd import weights as W
    from emojivoice_amd.hifigan import AttrDict, Generator, v1
    from emojivoice_amd.matcha_tts import MatchaTTS
    from emojivoice_amd.pipeline import BatchPipeline

    rank, world, local = D.init_from_env()
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    B, T, n_ode = args.batch, args.frames, args.ode_steps
    assert T % 4 == 0
    sd = W.synthetic_matcha_state()
    voc_sd = W.synthetic_hifigan_state()
    model = MatchaTTS(sd, device=device)
    voc = Generator(AttrDict(v1)).to(device)
    voc.load_state_dict(voc_sd)
    lo, hi = D.shard_bounds(B * world, rank, world)
    mu, z, spk_ids, lengths = make_inputs(B * world, T, lo, hi, device)
    spk = model._sd["spk_emb.weight"][spk_ids]
    z = z * 0.667

    def step_local():
        dec = model.engine.cfm_decode(mu, lengths, spk, z, n_ode, model.mel_std, model.mel_mean)   # denormalised mel
        return voc(dec)

    # consecutive batches are software-pipelined on two HIP streams (emojivoice_amd/pipeline.py): CFM decode of batch i+1
    # on a high-priority stream while HiFi-GAN of batch i runs; --no-pipeline runs the two stages back to back instead
    pipe = None if args.no_pipeline else BatchPipeline(model, voc)

    def step():
        if pipe is None:
            wav = step_local()
            return D.all_gather_waveforms(wav) if world > 1 else wav
        wav = pipe.submit(mu, lengths, spk, z, n_ode)
        if world > 1:
            with torch.cuda.stream(pipe.vocoder_stream):
                wav = D.all_gather_waveforms(wav)
        return wav

    log(f"[bench] rank {rank}/{world}: weights loaded, B={B} T={T}; warmup {args.warmup} ...")
    for _ in range(args.warmup):
        tw = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log(f"[bench] warmup step {time.perf_counter() - tw:.3f} s")
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wav = step()
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    log(f"[bench] timed {args.steps} steps in {dt:.3f} s")
    audio_s_total = args.steps * B * world * T * HOP / SR
    value = audio_s_total / dt

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel family (fp32-MFMA implicit-GEMM conv), measured live with HIP events
        # recorded on the launch stream around every conv launch of one extra, untimed-for-`value` step
        for e in (model.engine, voc.engine):
            e.profile_enable(True)
        step_local()                      # rank-local: no collective outside the timed region
        torch.cuda.synchronize()
        ms_c, fl_c, n_c = model.engine.profile_read()
        ms_v, fl_v, n_v = voc.engine.profile_read()
        for e in (model.engine, voc.engine):
            e.profile_enable(False)
        conv_ms, conv_fl, conv_n = ms_c + ms_v, fl_c + fl_v, n_c + n_v
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12
        # HBM bytes per conv launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs, FETCH doubled per the gfx950 note in MI355X_MICROARCH.md); not measurable live
        traffic = None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "r01_conv_hbm_traffic_pmc.json")))
            if B == 64 and T == 516 and n_ode == 10:
                traffic = {"bytes_per_launch": round(tj["hbm_MB_per_launch"] * 1e6), "GB_per_step": round(tj["hbm_GB_per_step"], 1),
                           "source": "profiles/r01_conv_hbm_traffic_pmc.json"}
        except Exception:
            pass
        per_gpu = value / world
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "kernel": "conv_gemm_kernel + resblock_pair_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit GEMM)",
                    "launches_per_step": int(conv_n), "avg_launch_us": round(conv_ms * 1e3 / max(conv_n, 1), 2),
                    "alg_gflop_per_launch": round(conv_fl / max(conv_n, 1) / 1e9, 3),
                    "conv_ms_per_step": round(conv_ms, 2), "conv_ms_cfm": round(ms_c, 2), "conv_ms_hifigan": round(ms_v, 2),
                    "tflops_cfm_convs": round(fl_c / (ms_c * 1e-3) / 1e12, 2), "tflops_hifigan_convs": round(fl_v / (ms_v * 1e-3) / 1e12, 2)}
        path_roof = {"fp32_frac": round(per_gpu * ALG_FLOP_PER_AUDIO_S / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
                     "hbm_frac": round(per_gpu * ALG_BYTES_PER_AUDIO_S / (PEAK_HBM_GBS * 1e9), 4),
                     "note": "whole-path fractions from SURVEY §8(d) per-audio-second work; the fp32 MFMA roof binds"}
        # latency of ONE batch through both stages back to back (what a single request sees; `value` is throughput)
        torch.cuda.synchronize()
        tl = time.perf_counter()
        step_local()
        torch.cuda.synchronize()
        batch_latency_ms = (time.perf_counter() - tl) * 1e3
        # PCIe-inclusive rate (never `value`): the same step plus the D2H copy of the waveform block into pinned host memory
        pcie = None
        try:
            host_wav = torch.empty((B,) + tuple(wav.shape[1:]), dtype=torch.float32, pin_memory=True)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            host_wav.copy_(step_local(), non_blocking=True)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - tp
            td = time.perf_counter()
            host_wav.copy_(wav[:B], non_blocking=True)
            torch.cuda.synchronize()
            pcie = {"audio_s_per_s_per_gpu": round(B * T * HOP / SR / dtp, 2), "ms_per_step": round(dtp * 1e3, 2),
                    "d2h_ms": round((time.perf_counter() - td) * 1e3, 3), "d2h_MB": round(host_wav.numel() * 4 / 1e6, 1)}
            del host_wav
        except Exception as ex:  # pinned allocation can be refused on small hosts: the field stays null
            log(f"[bench] pcie-inclusive probe skipped: {ex}")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            s = min(args.cpu_sample, B)
            cpu, ref_mel, ref_wav = cpu_baseline(sd, voc_sd, mu[:s].cpu(), z[:s].cpu(), spk[:s].cpu(), n_ode)
            # parity of the benchmarked configuration itself: the same rows decoded as their own batch on the GPU
            dec_s = model.engine.cfm_decode(mu[:s], lengths[:s], spk[:s], z[:s], n_ode, model.mel_std, model.mel_mean)
            wav_s = voc(dec_s)
            cpu["parity_mel_linf"] = float((dec_s.cpu() - ref_mel).abs().max())
            cpu["parity_wav_rms"] = float((wav_s.cpu() - ref_wav).pow(2).mean().sqrt())
        out = {
            "metric": "audio_seconds_per_second", "value": round(value, 2), "unit": "audio_s/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config2: batch {B} x {T}-frame (5.99 s) utterances per GPU, {n_ode} Euler steps + HiFi-GAN V1, 22.05 kHz",
                       "global_batch": B * world, "frames": T, "ode_steps": n_ode, "parallelism": f"dp{world}",
                       "collective": "all_gather(waveforms)" if world > 1 else "none",
                       "batch_pipeline": "off" if pipe is None else "cfm(i+1) || hifigan(i) on two streams"},
            "per_gpu_audio_s_per_s": round(per_gpu, 2), "rtf": round(1.0 / per_gpu, 6), "x_realtime_per_gpu": round(per_gpu, 1),
            "batch_latency_ms": round(batch_latency_ms, 2),
            "roofline": roofline, "path_roofline": path_roof, "cpu_baseline": cpu, "pcie_inclusive": pcie,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
