import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["XZAMD_TIMING"] = "1"
import torch, xz_amd, _oracle as o
enc = xz_amd.Encoder()
for name, data in (("lorem", o.corpus_lorem(229001)), ("text", xz_amd.corpus_text(1 << 20, seed=1000).tobytes()[:229001])):
    for preset in (1,):
        opts = xz_amd.preset_options(preset, span_size=xz_amd.SPAN_WHOLE_BLOCK)
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        print(name, preset, flush=True)
        out, _ = enc.encode(t, opts=opts, block_size=1 << 20)
        print("  enc ms", enc.stats().ms_encode, "out", out.numel(), flush=True)
