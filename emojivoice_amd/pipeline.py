"""Software pipeline over consecutive batches (no reference counterpart: the reference synthesises one utterance at a time,
feel_me.py:181-200).

The two stages of the path have opposite shapes on an MI355X: the CFM decode is ~600 short launches per batch that leave
the matrix pipes about half idle (single-round grids), HiFi-GAN is ~60 launches of thousands of workgroups each.  Run as
two stages on two HIP streams — the decode of batch i+1 on a high-priority stream, the vocoder of batch i on a normal one —
the decode's workgroups fill the gaps of the vocoder's grids: +5 % throughput at batch 64 (207 -> 197 ms per batch),
results bit-identical, per-batch latency unchanged.  Each stage owns its own native handle (workspace), so the stages never
touch the same buffers; the only hand-over is the mel tensor, ordered by an event.
"""
from __future__ import annotations

import torch


class BatchPipeline:
    """``submit()`` enqueues one batch (CFM decode -> HiFi-GAN) and returns the waveform tensor, valid on
    ``vocoder_stream``; ``synchronize()`` (or a wait on ``last_event``) makes it visible to the caller."""

    def __init__(self, model, vocoder):
        self.model, self.vocoder = model, vocoder
        dev = model.device
        import os
        pd, pv = (int(v) for v in os.environ.get("EV_PIPE_PRIO", "-1,0").split(","))   # A/B: stream priorities (decode, vocoder)
        self.decode_stream = torch.cuda.Stream(device=dev, priority=pd)
        self.vocoder_stream = torch.cuda.Stream(device=dev, priority=pv)
        self.last_event = None
        # the vocoder stays on ONE stream here: its small-call fan-out over three streams first drains the caller's stream on
        # the host (emojivoice.h, ev_hifigan), which would stall this pipeline's enqueue-ahead; the decode stream fills the gaps
        # The limit is the handle's: a vocoder that also serves single requests (EmojiTTS, to_waveform) gets its fan-out back
        # from close() / the context-manager exit.
        vocoder._sync_engine()
        # The limit is saved ONCE per engine, by the first pipeline that takes it away, and given back when the last one closes: a second
        # pipeline built on the same vocoder before the first was closed used to save the 0 the first one had set and "restore" that — the
        # fan-out was lost for good.
        eng = vocoder.engine
        if eng.pipeline_owner is None:
            eng.pipeline_owner = {"holders": 0, "saved": eng.mrf_streams_max}
            eng.set_mrf_streams_max(0)
        eng.pipeline_owner["holders"] += 1
        self._holding = True

    def close(self) -> None:
        """Drain both streams and give the vocoder engine its small-call three-stream fan-out limit back."""
        self.synchronize()
        eng = self.vocoder.engine
        if self._holding and eng is not None and eng.pipeline_owner is not None:
            self._holding = False
            eng.pipeline_owner["holders"] -= 1
            if eng.pipeline_owner["holders"] <= 0:
                if eng.h:
                    eng.set_mrf_streams_max(eng.pipeline_owner["saved"])
                eng.pipeline_owner = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def submit(self, mu, lengths, spk, z, n_timesteps: int, return_mel: bool = False):
        """mu / z: (B, n_feats, Tp) normalised encoder output and temperature-scaled noise (flow_matching.py:32-50),
        lengths (B,), spk (B, spk_emb_dim).  Same arithmetic as ``MatchaTTS.decode`` + ``Generator.forward``.
        ``return_mel``: also hand back the denormalised mel the vocoder consumed (valid on ``vocoder_stream`` like the waveform)."""
        cur = torch.cuda.current_stream(self.model.device)
        self.decode_stream.wait_stream(cur)               # inputs produced on the caller's stream
        with torch.cuda.stream(self.decode_stream):
            mel = self.model.engine.cfm_decode(mu, lengths, spk, z, n_timesteps, self.model.mel_std, self.model.mel_mean)
            ready = torch.cuda.Event()
            ready.record(self.decode_stream)
        with torch.cuda.stream(self.vocoder_stream):
            self.vocoder_stream.wait_event(ready)
            mel.record_stream(self.vocoder_stream)
            wav = self.vocoder(mel)
            self.last_event = torch.cuda.Event()
            self.last_event.record(self.vocoder_stream)
        return (wav, mel) if return_mel else wav

    def synchronize(self) -> None:
        self.decode_stream.synchronize()
        self.vocoder_stream.synchronize()


class PipelineGroup:
    """Several ``BatchPipeline``s in flight, consecutive batches submitted to them in turn.  Each needs its own (model, vocoder)
    pair — a native handle owns one workspace and serves one stream user at a time.  Two pipelines keep two vocoders and up to two
    decodes co-scheduled (bench.py: +1 % throughput at batch 64 over one pipeline; a third adds nothing); results are those of
    the serial calls, bit for bit."""

    def __init__(self, pairs):
        self.pipes = [BatchPipeline(m, v) for m, v in pairs]
        self._turn = 0

    @property
    def last(self) -> BatchPipeline:
        """The pipeline the most recent ``submit`` went to (its ``vocoder_stream`` orders the returned tensors)."""
        return self.pipes[(self._turn - 1) % len(self.pipes)]

    def submit(self, mu, lengths, spk, z, n_timesteps: int, return_mel: bool = False):
        p = self.pipes[self._turn % len(self.pipes)]
        self._turn += 1
        return p.submit(mu, lengths, spk, z, n_timesteps, return_mel=return_mel)

    def synchronize(self) -> None:
        for p in self.pipes:
            p.synchronize()

    def close(self) -> None:
        for p in self.pipes:
            p.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
