#!/bin/bash
# Re-assemble DESIGN.md / NOTES.md from tools/docs/*.md and the committed bench line.
cd "$(dirname "$0")/../.."
python tools/docs/assemble.py profiles/r04_bench_default.json \
  "AMAX_NUMBERS=3-tap layers −11 % (6 × 256→256 at 268 k rows: 2.43 → 2.15 ms), HiFi-GAN 62.5 → 61.4 ms, headline 4589 → 4628–4643 audio-s/s on one box (\`profiles/r04_amax_ab.txt\`)" \
  "CONFIG5FULL=p50 10.1 ms, p99 12.3 ms, mean 9.8 ms per utterance of 5.9 s on average (≈ 600× real time on one stream); round 3: 10.1 / 12.3 / 9.8 — unchanged, the batch-1 path is a chain of ≈ 1100 dependent launches"
