"""Timing of the STFT edge (pbbss_stft / pbbss_istft) on one CHiME-sized utterance and a
batch, against the NumPy oracle on the host.  Not part of bench.py (the headline metric is
the EM loop); reports achieved HBM bandwidth from the algorithmic bytes
(samples in + bins out, and back)."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--channels', type=int, default=6)
    ap.add_argument('--samples', type=int, default=128000)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--shift', type=int, default=256)
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--no-cpu', action='store_true')
    a = ap.parse_args()
    import torch
    from pb_bss_amd.transform import stft, istft
    rng = np.random.default_rng(0)
    for batch in (1, 32):
        C = a.channels * batch
        x = torch.from_numpy(rng.standard_normal((C, a.samples)).astype(np.float32)).cuda()

        def timed(fn):
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
            return out, e0.elapsed_time(e1) / a.reps * 1e3

        X, us_f = timed(lambda: stft(x, a.size, a.shift, dtype=np.complex64))
        _, us_l = timed(lambda: stft(x.reshape(batch, a.channels, -1)[0], a.size, a.shift,
                                     layout='f t d', dtype=np.complex64))
        y, us_i = timed(lambda: istft(X, a.size, a.shift))
        T, F = X.shape[-2:]
        bytes_f = C * a.samples * 4 + C * T * F * 8
        bytes_i = C * T * F * 8 + C * y.shape[-1] * 8
        err = float((y[..., :a.samples] - x.double()).abs().max())
        print(f'channels={C} samples={a.samples} size={a.size} shift={a.shift} frames={T}: '
              f'stft {us_f:.1f} us ({bytes_f / us_f / 1e3:.0f} GB/s), '
              f"stft 'f t d' one utterance {us_l:.1f} us, "
              f'istft {us_i:.1f} us ({bytes_i / us_i / 1e3:.0f} GB/s), round trip max err {err:.1e}')
    if not a.no_cpu:
        from oracle import stft as o
        xh = rng.standard_normal((a.channels, a.samples))
        t0 = time.perf_counter()
        Xh = o.stft(xh, a.size, a.shift)
        t1 = time.perf_counter()
        o.istft(Xh, a.size, a.shift)
        t2 = time.perf_counter()
        print(f'numpy oracle, {a.channels} channels: stft {(t1 - t0) * 1e3:.1f} ms, '
              f'istft {(t2 - t1) * 1e3:.1f} ms')


if __name__ == '__main__':
    main()
