"""Replay one saved 44 MHz capture (.npy, int16 [n, 2]) through the 802.11b receiver on the GPU and through the compiled reference graph, and
print both event lists.  tools/stress_parity_11b.py saves the capture of a mismatch under gpurun_out/.
    python tools/replay_11b.py capture.npy"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

if __name__ == "__main__":
    import torch, sora_amd
    from oracle.pyoracle import ReferenceGraph
    c = np.load(sys.argv[1])
    g = ReferenceGraph()
    ev = g.rx11b(c, max_frames=64) if g.available() else []
    rx = sora_amd.Rx11b(1, len(c), max_frames_per_capture=64)
    rx.process_dev(torch.from_numpy(c).cuda(), sora_amd.Rx.captures([(0, len(c), 0)]))
    got = rx.results(); rx.close()
    print("reference:")
    for e in ev: print("  end %7d  code %08x  rate %5d  len %4d" % (e["sample_index"], e["error_code"], e["rate_kbps"], e["length"]))
    print("gpu:")
    for r in got: print("  end %7d  code %08x  rate %5d  len %4d" % (r["end_sample"], r["error_code"], r["rate_kbps"], r["length"]))
