"""Oracle: the CTC head of the multi-task models (numpy float64, explicit loops).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the arithmetic lives in
TensorFlow 1.x (absent here); this file restates the published algorithms the reference calls and
tests/test_oracle_ctc.py cross-checks it against independent implementations (torch's
``ctc_loss`` + autograd, exhaustive enumeration of alignments).

Reference call sites (av_speech_inpainting/models.py):
* ``tf.nn.ctc_loss(sparse_labels, tm_logits, sequence_length, preprocess_collapse_repeated=False,
  ctc_merge_repeated=True, time_major=True)`` followed by ``reduce_mean`` (:1949-1952, :1641-1644);
* ``tf.contrib.keras.backend.ctc_label_dense_to_sparse(labels, labels_lengths)`` (:1494, :1760);
* ``tf.nn.ctc_beam_search_decoder(tm_logits, sequence_lengths, beam_width=20)`` -- top_paths=1,
  merge_repeated=True are the TF 1.x defaults (:1937-1940);
* ``tf.edit_distance(sparse_decoding, sparse_labels)`` -- normalize=True default (:2028).

Semantics restated (TF 1.13-1.15, core/util/ctc/ctc_loss_calculator.*, ctc_beam_search.h):
the blank label is ``num_classes - 1``; the loss takes UN-normalised logits and applies a softmax
per frame; loss[b] = -log p(labels_b | logits_b[:seq_len_b]); frames t >= seq_len_b get zero
gradient; d loss / d logit[t, k] = softmax[t, k] - (1 / p) sum_{s: l'_s = k} alpha_t(s) beta_t(s) / y_t(l'_s).
"""
import numpy as np

NEG_INF = -np.inf


def log_softmax(x):
    x = np.asarray(x, dtype=np.float64)
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))


def _lse(*xs):
    m = max(xs)
    if m == NEG_INF:
        return NEG_INF
    return m + np.log(sum(np.exp(x - m) for x in xs))


def dense_to_sparse(labels, labels_lengths):
    """ctc_label_dense_to_sparse: row b keeps its first labels_lengths[b] entries (cast to int32)."""
    labels = np.asarray(labels)
    return [labels[b, :int(n)].astype(np.int32) for b, n in enumerate(labels_lengths)]


def ctc_loss_one(logits, label, want_grad=True):
    """logits [T, C] (the frames inside the utterance), label int[L] -> (loss, d loss / d logits [T, C]).

    An infeasible labelling (needs more frames than T) gives loss = +inf and a zero gradient."""
    logits = np.asarray(logits, dtype=np.float64)
    T, C = logits.shape
    blank = C - 1
    label = [int(v) for v in label]
    if any(v < 0 or v >= blank for v in label):
        raise ValueError("labels must lie in [0, num_classes - 1)")
    ext = [blank]
    for v in label:
        ext += [v, blank]
    S = len(ext)
    lp = log_softmax(logits)
    if T == 0:
        return (0.0 if not label else np.inf), np.zeros_like(logits)

    def can_skip(s):        # transition s-2 -> s
        return s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]
    alpha = np.full((T, S), NEG_INF)
    alpha[0, 0] = lp[0, blank]
    if S > 1:
        alpha[0, 1] = lp[0, ext[1]]
    for t in range(1, T):
        for s in range(S):
            terms = [alpha[t - 1, s]]
            if s >= 1:
                terms.append(alpha[t - 1, s - 1])
            if can_skip(s):
                terms.append(alpha[t - 1, s - 2])
            alpha[t, s] = _lse(*terms) + lp[t, ext[s]]
    log_p = _lse(alpha[T - 1, S - 1], alpha[T - 1, S - 2]) if S > 1 else alpha[T - 1, 0]
    if log_p == NEG_INF:
        return np.inf, np.zeros_like(logits)
    if not want_grad:
        return -log_p, None
    beta = np.full((T, S), NEG_INF)          # includes y_t, like alpha
    beta[T - 1, S - 1] = lp[T - 1, blank]
    if S > 1:
        beta[T - 1, S - 2] = lp[T - 1, ext[S - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(S):
            terms = [beta[t + 1, s]]
            if s + 1 < S:
                terms.append(beta[t + 1, s + 1])
            if s + 2 < S and can_skip(s + 2):
                terms.append(beta[t + 1, s + 2])
            beta[t, s] = _lse(*terms) + lp[t, ext[s]]
    grad = np.exp(lp)
    for t in range(T):
        for s in range(S):
            g = alpha[t, s] + beta[t, s] - lp[t, ext[s]] - log_p
            if g > NEG_INF:
                grad[t, ext[s]] -= np.exp(g)
    return -log_p, grad


def ctc_loss(logits, labels, labels_lengths, sequence_lengths, want_grad=True):
    """logits [B, T, C] batch-major; labels [B, Lmax]; -> (loss [B], d sum(loss) / d logits [B, T, C])."""
    logits = np.asarray(logits, dtype=np.float64)
    B = logits.shape[0]
    loss = np.zeros(B)
    grad = np.zeros_like(logits)
    for b, lab in enumerate(dense_to_sparse(labels, labels_lengths)):
        n = int(sequence_lengths[b])
        loss[b], g = ctc_loss_one(logits[b, :n], lab, want_grad)
        if want_grad:
            grad[b, :n] = g
    return loss, (grad if want_grad else None)


class _Beam:
    __slots__ = ('labels', 'o_blank', 'o_label', 'o_total', 'n_blank', 'n_label', 'n_total')

    def __init__(self, labels):
        self.labels = labels
        self.o_blank = self.o_label = self.o_total = NEG_INF
        self.n_blank = self.n_label = self.n_total = NEG_INF


def beam_search_one(logits, beam_width=20, merge_repeated=True):
    """TF's CTCBeamSearchDecoder (ctc_beam_search.h), top path only.  logits [T, C] -> (labels, log prob).

    Beam entries are label prefixes; per step: entries of the beam are extended in time (same
    prefix, via blank or a repeat of the last label, plus the mass flowing in from the parent if it
    is still in the beam), then every prefix whose previous total could still enter the beam
    spawns its children.  ``merge_repeated`` collapses consecutive equal labels of the OUTPUT."""
    logits = np.asarray(logits, dtype=np.float64)
    T, C = logits.shape
    blank = C - 1
    lp = log_softmax(logits) if T else logits
    root = _Beam(())
    root.n_blank = root.n_total = 0.0
    entries = {(): root}                     # prefix -> entry, the entries currently in the beam
    for t in range(T):
        branches = sorted(entries.values(), key=lambda e: -e.n_total)
        for b in branches:
            b.o_blank, b.o_label, b.o_total = b.n_blank, b.n_label, b.n_total
        leaves = {}
        for b in branches:
            if b.labels:
                parent = entries.get(b.labels[:-1])
                if parent is not None:
                    same = len(b.labels) >= 2 and b.labels[-1] == b.labels[-2]
                    b.n_label = _lse(b.n_label, parent.o_blank if same else parent.o_total)
                b.n_label += lp[t, b.labels[-1]]
            b.n_blank = b.o_total + lp[t, blank]
            b.n_total = _lse(b.n_blank, b.n_label)
            leaves[b.labels] = b

        def bottom():
            return min(leaves.values(), key=lambda e: e.n_total)

        def is_candidate(total):
            return total > NEG_INF and (len(leaves) < beam_width or total > bottom().n_total)
        for b in branches:
            if not is_candidate(b.o_total):
                continue
            for k in range(C - 1):
                key = b.labels + (k,)
                if key in leaves:
                    continue
                prev = b.o_blank if (b.labels and b.labels[-1] == k) else b.o_total
                c = _Beam(key)
                c.n_label = c.n_total = lp[t, k] + prev
                if is_candidate(c.n_total):
                    if len(leaves) == beam_width:
                        del leaves[bottom().labels]
                    leaves[key] = c
                elif key in entries:
                    # a prefix pushed out of the beam earlier in this step and found again as a child that
                    # does not make it back: TF deactivates the node (oldp.Reset()), so it spawns nothing
                    # when its own turn comes
                    entries[key].o_blank = entries[key].o_label = entries[key].o_total = NEG_INF
        entries = leaves
    best = max(entries.values(), key=lambda e: e.n_total)
    out = list(best.labels)
    if merge_repeated:
        out = [v for i, v in enumerate(out) if i == 0 or v != out[i - 1]]
    return out, best.n_total


def beam_search(logits, sequence_lengths, beam_width=20, merge_repeated=True):
    """-> (list of int label lists, log probabilities [B]); dense form: pad with -1 (sparse.to_dense default)."""
    outs, scores = [], []
    for b in range(len(sequence_lengths)):
        o, s = beam_search_one(np.asarray(logits)[b, :int(sequence_lengths[b])], beam_width, merge_repeated)
        outs.append(o)
        scores.append(s)
    return outs, np.array(scores)


def edit_distance(hyp, truth, normalize=True):
    """tf.edit_distance on one pair of sequences: Levenshtein distance, divided by len(truth)."""
    n, m = len(hyp), len(truth)
    d = list(range(m + 1))
    for i in range(1, n + 1):
        prev, d[0] = d[0], i
        for j in range(1, m + 1):
            cur = min(d[j] + 1, d[j - 1] + 1, prev + (hyp[i - 1] != truth[j - 1]))
            prev, d[j] = d[j], cur
    dist = float(d[m])
    if normalize:
        if m == 0:
            return np.inf if n else 0.0
        return dist / m
    return dist
