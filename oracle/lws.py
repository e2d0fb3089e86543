"""Oracle: LWS ("local weighted sums") phase reconstruction as the reference's ``infer`` uses it (numpy, float64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED -- and more so than the rest: the reference calls
the third-party ``lws`` package (``lws.lws(384, 192, fftsize=512, mode='speech')``, inference.py:119, un-pinned and not
even listed in requirements.txt), which is neither vendored in /root/reference nor installable here.  What follows
restates the PUBLISHED algorithm

    J. Le Roux, H. Kameoka, N. Ono, S. Sagayama, "Fast signal reconstruction from magnitude STFT spectrogram based on
    spectrogram consistency", DAFx-10 (2010)                                                    -- batch LWS
    J. Le Roux, H. Kameoka, N. Ono, S. Sagayama, "Phase initialization schemes for faster spectrogram-consistency-
    based signal reconstruction", ASJ autumn meeting (2010)                     -- "no future" and online LWS

and the package's documented interface (constructor arguments, ``mode='speech'`` defaults, ``stft`` / ``run_lws`` /
``istft``) -- and the reference's own stitching code around it (inference.py:141-154), which IS available and is
followed literally.  Every choice the publications leave open is listed under "Conventions" and is shared by the
HIP implementation (csrc/lws.hip), so the parity tests compare two implementations of the same written definition.

The algorithm.  A complex array W[m, k] is the STFT of some signal iff it is a fixed point of
F = STFT o iSTFT (its "consistency").  With analysis window w, synthesis window s (perfect reconstruction), frame
length N and shift R, F is a local operator:

    F(W)[m, k] = sum_{q = -(Q-1)}^{Q-1}  sum_p  alpha_q(p) . exp(-2 pi j (k + p) q R / N) . W[m + q, k + p]
    alpha_q(p) = 1/N sum_n w(n) s(n - q R) exp(+2 pi j p n / N)

(derived in ``consistency_weights``; checked numerically against STFT(iSTFT(W)) in tests/test_oracle_lws.py).
alpha_q(p) decays fast in |p|, so the sum is truncated to |p| <= L (L = 5).  LWS keeps the magnitudes A and updates
phases bin by bin: S[m, k] <- A[m, k] . t / |t| with t the truncated sum WITHOUT its centre term (alpha_0(0) is real
and positive: it only adds inertia).  Bins are visited in raster order IN PLACE, and only bins whose magnitude
exceeds a threshold that decays over the iterations (large bins first: the "sparsity" speed-up of the paper).

Conventions (unpinned choices)
* windows: ``lws(fsize, fshift)`` builds the sqrt of a SYMMETRIC Hann window (package default ``symmetric_win=True``)
  and the synthesis window ``awin / sum_q awin(n + q R)^2``; ``fftsize`` > fsize zero-pads both windows symmetrically
  to the FFT length, which then is the frame length N of the theory (384 -> 512: 64 zeros either side).
* ``stft``: 'perfectrec' padding -- N - R zeros in front and behind, then zeros up to a whole number of frames;
  ``istft`` removes the front padding.  For 48,000 samples: M = 252 frames.
* ``mode='speech'``: look_ahead 3, one "no future" pass (alpha 1), one online pass (alpha 1), 100 batch iterations
  with thresholds ``100 exp(-0.1 j)`` (batch_alpha 100, beta 0.1, gamma 1), all relative to the MEAN magnitude of
  the spectrogram.
* "no future" pass: frames in order, phases of frame m from the already processed frames only (rows q < 0).
* online pass: frames in order, frame m updated in place from the full neighbourhood; frames m + 1 .. m + look_ahead
  hold their "no future" estimates -- which is simply one in-place sweep after the "no future" pass.
* batch iterations: in-place raster sweeps over all frames (neighbour rows outside the spectrogram are zero).
* spectra are conjugate-symmetric: bins k < 0 and k > N/2 of the neighbourhood are mirror images, refreshed as soon
  as their source bin changes.
"""
import numpy as np


def hann_symmetric(n):
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(n) / (n - 1)))


def synthesis_window(awin, fshift):
    """Perfect-reconstruction synthesis window for ``awin`` at shift ``fshift``: awin / sum_q awin(n + q R)^2."""
    n = len(awin)
    Q = -(-n // fshift)
    w2 = np.concatenate([awin * awin, np.zeros(Q * fshift - n)]).reshape(Q, fshift).sum(axis=0)
    den = np.tile(w2, Q)[:n]
    if den.min() <= 0:
        raise ValueError("the window overlap normaliser is not strictly positive")
    return awin / den


class LWS(object):
    """Restatement of ``lws.lws(awin_or_fsize, fshift, L=5, look_ahead=3, ..., mode=None, fftsize=None)``."""

    def __init__(self, fsize, fshift, L=5, look_ahead=3, nofuture_iterations=0, nofuture_alpha=1.0,
                 online_iterations=0, online_alpha=1.0, batch_iterations=100, batch_alpha=100.0, batch_beta=0.1,
                 batch_gamma=1.0, mode=None, fftsize=None):
        if mode == 'speech':
            look_ahead, nofuture_iterations, nofuture_alpha = 3, 1, 1.0
            online_iterations, online_alpha = 1, 1.0
            batch_iterations, batch_alpha = 100, 100.0
        elif mode == 'music':
            look_ahead, nofuture_iterations, nofuture_alpha = 3, 1, 1.0
            online_iterations, online_alpha = 10, 1.0
            batch_iterations, batch_alpha = 100, 100.0
        awin = np.sqrt(hann_symmetric(int(fsize)))
        swin = synthesis_window(awin, fshift)
        N = int(fftsize) if fftsize else int(fsize)
        if N < fsize:
            raise ValueError("fftsize must not be smaller than the window")
        lo = (N - fsize) // 2
        self.awin = np.pad(awin, (lo, N - fsize - lo))
        self.swin = np.pad(swin, (lo, N - fsize - lo))
        self.N, self.R, self.L = N, int(fshift), int(L)
        self.Q = -(-N // self.R)
        self.look_ahead = look_ahead
        self.nofuture_iterations, self.nofuture_alpha = nofuture_iterations, nofuture_alpha
        self.online_iterations, self.online_alpha = online_iterations, online_alpha
        self.batch_iterations = batch_iterations
        self.batch_alpha, self.batch_beta, self.batch_gamma = batch_alpha, batch_beta, batch_gamma
        self.alpha = consistency_weights(self.awin, self.swin, self.R, self.L)

    # ------------------------------------------------------------------------------------------ transforms
    def num_frames(self, n):
        pad = self.N - self.R
        return max(1, -(-(n + 2 * pad - self.N) // self.R) + 1)

    def stft(self, x):
        """[n] -> complex [M, N/2 + 1] ('perfectrec' padding)."""
        x = np.asarray(x, dtype=np.float64)
        pad = self.N - self.R
        M = self.num_frames(len(x))
        xp = np.zeros((M - 1) * self.R + self.N)
        xp[pad:pad + len(x)] = x
        idx = np.arange(M)[:, None] * self.R + np.arange(self.N)[None, :]
        return np.fft.rfft(xp[idx] * self.awin[None, :], axis=1)

    def istft(self, S):
        """complex [M, N/2 + 1] -> [(M - 1) R + N - 2 (N - R)] (the front and back padding removed)."""
        M = S.shape[0]
        fr = np.fft.irfft(S, n=self.N, axis=1) * self.swin[None, :]
        y = np.zeros((M - 1) * self.R + self.N)
        for m in range(M):
            y[m * self.R:m * self.R + self.N] += fr[m]
        pad = self.N - self.R
        return y[pad:len(y) - pad]

    # ------------------------------------------------------------------------------------------ the iterations
    def thresholds(self, stage):
        if stage == 'nofuture':
            return [self.nofuture_alpha] * self.nofuture_iterations
        if stage == 'online':
            return [self.online_alpha] * self.online_iterations
        return [self.batch_alpha * np.exp(-self.batch_beta * j ** self.batch_gamma) for j in range(self.batch_iterations)]

    def sweep_schedule(self):
        """[(past_only, relative threshold)] for one run_lws call."""
        return ([(True, t) for t in self.thresholds('nofuture')] + [(False, t) for t in self.thresholds('online')]
                + [(False, t) for t in self.thresholds('batch')])

    def run_lws(self, S0, schedule=None, trace=None):
        """complex [M, N/2 + 1] (magnitudes to keep, phases to start from) -> complex [M, N/2 + 1]."""
        S = np.array(S0, dtype=np.complex128)
        amp = np.abs(S)
        mean_amp = amp.mean()
        for past_only, rel in (self.sweep_schedule() if schedule is None else schedule):
            sweep(S, amp, rel * mean_amp, self.alpha, self.N, self.R, self.L, past_only)
            if trace is not None:
                trace.append(self.inconsistency(S))
        return S

    def inconsistency(self, S):
        """|| STFT(iSTFT(S)) - S ||^2 / || S ||^2 with this object's own transforms (exact, no truncation)."""
        y = np.fft.irfft(S, n=self.N, axis=1) * self.swin[None, :]
        M = S.shape[0]
        sig = np.zeros((M - 1) * self.R + self.N)
        for m in range(M):
            sig[m * self.R:m * self.R + self.N] += y[m]
        idx = np.arange(M)[:, None] * self.R + np.arange(self.N)[None, :]
        F = np.fft.rfft(sig[idx] * self.awin[None, :], axis=1)
        return float((np.abs(F - S) ** 2).sum() / (np.abs(S) ** 2).sum())


def consistency_weights(awin, swin, R, L):
    """alpha_q(p) = 1/N sum_n w(n) s(n - q R) e^{+2 pi j p n / N} for |q| <= Q - 1, |p| <= L -> complex [2Q-1, 2L+1].

    Derivation: X[m,k] = sum_n w(n) x(n + mR) e^{-2 pi j k n / N};  x(t) = sum_m' s(t - m'R) 1/N sum_k' X[m',k'] e^{2 pi j k'(t - m'R)/N}.
    Substituting, with q = m' - m and p = k' - k:
        F(X)[m,k] = sum_q sum_p X[m+q, k+p] e^{-2 pi j (k+p) q R / N} alpha_q(p)."""
    N = len(awin)
    Q = -(-N // R)
    n = np.arange(N)
    out = np.zeros((2 * Q - 1, 2 * L + 1), dtype=np.complex128)
    for q in range(-(Q - 1), Q):
        sh = np.zeros(N)
        src = n - q * R
        ok = (src >= 0) & (src < N)
        sh[ok] = swin[src[ok]]
        prod = awin * sh
        for p in range(-L, L + 1):
            out[q + Q - 1, p + L] = (prod * np.exp(2j * np.pi * p * n / N)).sum() / N
    return out


def extend(row, L):
    """One frame [N/2 + 1] -> [N/2 + 1 + 2L]: conjugate mirror images below DC and above Nyquist."""
    return np.concatenate([np.conj(row[L:0:-1]), row, np.conj(row[-2:-L - 2:-1])])


def sweep(S, amp, thr, alpha, N, R, L, past_only):
    """One in-place raster sweep over all frames and bins (strictly sequential: a bin sees every earlier update of
    this sweep, including through the mirror images)."""
    M, K = S.shape
    Q = (alpha.shape[0] + 1) // 2
    omega = np.exp(-2j * np.pi * R / N)
    zero = np.zeros(K + 2 * L, dtype=np.complex128)
    pidx = np.arange(-L, L + 1)
    for m in range(M):
        if not (amp[m] > thr).any():
            continue
        ext = {q: (extend(S[m + q], L) if 0 <= m + q < M else zero) for q in range(-(Q - 1), Q)}
        rows = [q for q in range(-(Q - 1), Q) if (q < 0 if past_only else True)]
        for k in range(K):
            if not amp[m, k] > thr:
                continue
            t = 0j
            for q in rows:
                w = alpha[q + Q - 1] * omega ** (((k + pidx) * q) % N)
                if q == 0:
                    w = w.copy()
                    w[L] = 0.0                      # no centre term
                t += (w * ext[q][k:k + 2 * L + 1]).sum()
            a = abs(t)
            if a > 0:
                S[m, k] = amp[m, k] * t / a
                # refresh the current row's extension (bin k and its mirror image)
                e = ext[0]
                e[k + L] = S[m, k]
                if 1 <= k <= L:
                    e[L - k] = np.conj(S[m, k])
                if K - 1 - L <= k <= K - 2:
                    e[L + 2 * (K - 1) - k] = np.conj(S[m, k])


def refine_enhanced(lws, enhanced, mask):
    """inference.py:141-154, literally: the gap phases of one enhanced waveform are replaced by LWS estimates.

    enhanced [n] (output of enhanced_sources: masked target phase, zero phase inside gaps), mask [T, F] -> [n']."""
    stft = lws.stft(enhanced)
    mask_adj = np.zeros(stft.shape)
    mask_adj[: mask.shape[0], : mask.shape[1]] = mask
    mag_spec = np.abs(stft)
    ang_spec = np.angle(stft) * mask_adj
    rec_stft = lws.run_lws(mag_spec * np.exp(1j * ang_spec))
    rec_mag = np.abs(rec_stft)
    rec_ang = np.angle(rec_stft)
    rec_ang_adj = ang_spec + rec_ang * (1 - mask_adj)
    rec_stft_adj = rec_mag * np.exp(1j * rec_ang_adj)
    return lws.istft(rec_stft_adj)
