"""power check of tests/test_input_pipeline.py::test_ring_slots_are_not_refilled_before_their_upload: the same loop with the
upload events thrown away (slots refilled as soon as the DataLoader machinery allows) must show corrupted frames."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "monocon-pytorch_amd"), REPO, os.path.join(REPO, "tests")]
import torch
from test_input_pipeline import _TaggedFrames

if __name__ == "__main__":
    from hipmonocon.feed import DevicePrefetcher, RingLoader
    for broken in (False, True):
        ds = _TaggedFrames(600 * 8, (3, 96, 512))
        rl = RingLoader(ds, batch_size=8, num_workers=4, shuffle=True, collate_fn=ds.collate_fn)
        if broken:
            rl.note_upload = lambda ev: None
        pf = DevicePrefetcher(rl, "cuda:0")
        ballast_host = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
        ballast_dev = torch.empty_like(ballast_host, device="cuda")
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        for b in pf:
            with torch.cuda.stream(pf.copy_stream):
                ballast_dev.copy_(ballast_host, non_blocking=True)
            want = torch.tensor(b["img_metas"]["sample_idx"], dtype=torch.float32).cuda(non_blocking=True)
            bad += (b["img"][:, 0, 0, 0] != want).sum() + (b["img"][:, -1, -1, -1] != want).sum()
        print("events %s: %d corrupted frame ends" % ("thrown away" if broken else "honoured", int(bad)), flush=True)
        rl.close()
