"""Stage timing of one batch (default 1368 MiB of the bench text, preset 6); with XZ_AMD_LIB=xz_amd/libxz_amd_timing.so and
XZAMD_TIMING=1 the kernels also print their in-kernel cycle breakdown."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xz_amd


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1368
    preset = int(sys.argv[2], 0) if len(sys.argv) > 2 else 6
    corpus = sys.argv[3] if len(sys.argv) > 3 else "text"
    n = mib << 20
    data = xz_amd.corpus_text(n) if corpus == "text" else xz_amd.corpus_tar(n)
    enc = xz_amd.Encoder(0)
    t = torch.from_numpy(data).to("cuda:0")
    opts = xz_amd.preset_options(preset)
    if len(sys.argv) > 4:
        opts.span_cost = int(sys.argv[4])
    if len(sys.argv) > 5:
        opts.enc_span_bits = int(sys.argv[5])
    for it in range(2):
        out, _ = enc.encode(t, opts=opts)
        st = enc.stats()
        print(f"iter {it}: out {st.out_bytes} ratio {st.out_bytes / n:.4f} pieces {st.spans} enc_spans {st.enc_spans} | ms chains {st.ms_chains:.0f} "
              f"find {st.ms_find:.0f} plan {st.ms_plan:.1f} seed {st.ms_seed:.0f} parse {st.ms_parse:.0f} code {st.ms_code:.0f} "
              f"crc {st.ms_crc:.0f} assemble {st.ms_assemble:.0f} total {st.ms_total:.0f}", flush=True)


if __name__ == "__main__":
    main()
