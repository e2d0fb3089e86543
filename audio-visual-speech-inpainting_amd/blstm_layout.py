"""Parameter layouts of the stacked BLSTM: the reference's variables vs. the kernels' packed form.

Two flat float32 buffers describe the same numbers:

* the REFERENCE layout -- the TF variables of ``StackedBLSTMModel`` in creation order
  (reference models.py:106-121): per layer and direction ``kernel [(D_l+H), 4H]``
  (gate column blocks i, j, f, o) and ``bias [4H]``, then ``logits/weights [2H, F]`` and
  ``logits/biases [F]``.  Checkpoints, the optimiser and the data-parallel all-reduce use this;
* the PACKED layout the gfx950 kernels read (avsi_hip.h): hidden size padded 250 -> 256, gate
  columns interleaved per 32-unit wave slice, the recurrent kernel in MFMA fragment order.

``ParamLayout`` holds the index maps between them; moving data is one gather each way.
"""
import numpy as np

HP = 256            # padded hidden size of the recurrent kernel
GP = 4 * HP         # packed gate columns per direction
WH_FLOATS = 8 * 32 * 4 * 64 * 4   # packed recurrent kernel per direction (= HP * GP)


def round_up(a, b):
    return -(-a // b) * b


def input_pitch(dim):
    """Padded width of a network-input row: a multiple of 16 so that every input-projection GEMM
    qualifies for the LDS-DMA kernel (reduction length % 16 == 0)."""
    return round_up(dim, 16)


def packed_gate_col(direction, gate, unit):
    """Column of (direction, gate, hidden unit) inside the 2048-wide packed gate row."""
    return direction * GP + (unit // 32) * 128 + gate * 32 + unit % 32


_INDEX_CACHE = {}      # constructor arguments -> (pack_index, grad_index); shared between layouts, never written


class ParamLayout:
    """``side = (layer, dim)`` declares a per-utterance side input (a speaker embedding) that the
    reference concatenates, tiled over time, to the input of BLSTM layer ``layer``
    (models.py:848-851,1205-1209 for layer 0; :905-906,1247-1249 after ``integration_layer``
    layers).  Its rows of the TF kernel sit between the layer's input rows and the recurrent rows;
    the kernels never see the tiled tensor -- the rows are packed apart as ``we`` [dim_p, 2048] and
    enter the gate pre-activations as a per-utterance bias (see StackedBLSTMModel._forward).

    ``mlp = width`` adds the speaker-embedding network of StackedBLSTMSSNNModel
    (models.py:803-810): ``speaker_embedding/weights_1 [2F, width]``, ``biases_1``, ``weights_2 [width,
    width]``, ``biases_2``, ``weights_3``, ``biases_3``.  Packed: ``mw1a`` / ``mw1b`` = the feature and
    the delta-feature halves of weights_1 (rows padded to the model's time-major input pitch so the
    GEMM reads the input buffer in place), ``mw2``, ``mw3``, ``mb1..3``.

    ``asr = num_classes`` adds the second head of the multi-task CTC models (models.py:1903-1916):
    ``asr/weights [2H, num_classes]``, ``asr/biases``; the inpainting head keeps the internal names
    ``logits/*`` (its TF scope there is ``inpainting/``, see tf_checkpoint).  Packed, the two heads
    share ``pw`` / ``pb``: the asr columns start at ``asr_col`` = F rounded up to 4, so both heads
    are column windows of one matrix and the backward pass treats them as one projection."""

    def __init__(self, input_dim, net_dim=(250, 250, 250), audio_feat_dim=257, side=None, mlp=None, mlp_in_pitch=None,
                 asr=None):
        net_dim = tuple(int(h) for h in net_dim)
        # the C ABI says which stacks its kernels take (avsi_blstm_net_supported: any per-layer widths of 1 .. 256 units,
        # each layer padded to 256 of its own; the reference takes any width, models.py:95-99,107) -- the host layer has no rule of its own
        import ctypes
        from . import _lib
        rc = _lib.lib().avsi_blstm_net_supported((ctypes.c_int * len(net_dim))(*net_dim), len(net_dim)) if net_dim else -1
        if rc != _lib.AVSI_OK:
            raise _lib.AvsiError("net_dim = %r: %s (%d) -- the gfx950 BLSTM kernels take layer widths of 1 .. %d units"
                                 % (list(net_dim), _lib.lib().avsi_status_string(rc).decode(), rc, HP))
        self.input_dim = int(input_dim)
        # units per direction of every layer (the reference takes any, models.py:95-99,107; the kernels keep 256 per
        # direction, so every width is padded to HP with zero weights); H = the TOP layer's: what the projection reads
        self.Hs = net_dim
        self.H = H = net_dim[-1]
        self.num_layers = len(net_dim)
        self.F = F = int(audio_feat_dim)
        self.asr = int(asr) if asr else 0
        if self.asr and self.asr < 2:
            raise ValueError("the asr head needs at least one label and the blank, got %r classes" % (asr,))
        self.asr_col = round_up(F, 4) if self.asr else 0
        self.ldp = round_up(self.asr_col + self.asr, 4) if self.asr else round_up(F, 4)   # row pitch of the projection matrix
        self.side = None if side is None else (int(side[0]), int(side[1]))
        if self.side is not None and not (0 <= self.side[0] < self.num_layers and self.side[1] > 0):
            raise ValueError("side input must name an existing layer and a positive width, got %r" % (side,))
        self.side_p = round_up(self.side[1], 8) if self.side else 0
        self.mlp = int(mlp) if mlp else 0
        self.mlp_in_pitch = int(mlp_in_pitch or input_pitch(F))
        if self.mlp and (self.mlp % 8 or self.mlp_in_pitch < F):
            raise ValueError("speaker-embedding width must be a multiple of 8 and its input pitch >= %d" % F)

        # ---- reference layout
        self.ref_entries = []                          # (name, shape, offset)
        off = 0
        self.in_dims = []
        d = self.input_dim
        for li in range(self.num_layers):
            self.in_dims.append(d)
            e = self.side_dim(li)
            Hl = net_dim[li]
            for dname in ('fw', 'bw'):
                for vname, shape in (('kernel', (d + e + Hl, 4 * Hl)), ('bias', (4 * Hl,))):
                    self.ref_entries.append(('cell_%d/%s/%s' % (li, dname, vname), shape, off))
                    off += int(np.prod(shape))
            d = 2 * Hl
        self.ref_entries.append(('logits/weights', (2 * H, F), off))
        off += 2 * H * F
        self.ref_entries.append(('logits/biases', (F,), off))
        off += F
        if self.asr:
            self.ref_entries.append(('asr/weights', (2 * H, self.asr), off))
            off += 2 * H * self.asr
            self.ref_entries.append(('asr/biases', (self.asr,), off))
            off += self.asr
        if self.mlp:
            W = self.mlp
            for name, shape in (('weights_1', (2 * F, W)), ('biases_1', (W,)), ('weights_2', (W, W)), ('biases_2', (W,)),
                                ('weights_3', (W, W)), ('biases_3', (W,))):
                self.ref_entries.append(('speaker_embedding/' + name, shape, off))
                off += int(np.prod(shape))
        self.ref_size = off
        self._ref_off = {n: o for n, _, o in self.ref_entries}

        # ---- packed layout
        self.kp = [input_pitch(self.input_dim)] + [2 * HP] * (self.num_layers - 1)
        # A padded input column per layer (and of the projection's input) that the training pass sets to 1: its
        # weight row is zero, so the forward pass is unchanged, and its row of dW = X^T . dZ is the column sum of
        # dZ -- the bias gradient -- for free inside the weight-gradient GEMM.  -1: no padded column to spare
        # (input width a multiple of 16), that layer's bias gradient is a separate column sum.
        self.ones_col = [self.input_dim if self.kp[0] > self.input_dim else -1] + \
            [net_dim[li - 1] if net_dim[li - 1] < HP else -1 for li in range(1, self.num_layers)]
        self.ones_col_top = H if H < HP else -1
        self.packed = {}                               # name -> (offset, shape)
        poff = 0

        def alloc(name, shape):
            nonlocal poff
            self.packed[name] = (poff, shape)
            poff += round_up(int(np.prod(shape)), 64)
        for li in range(self.num_layers):
            alloc('wx%d' % li, (self.kp[li], 2 * GP))
            alloc('b%d' % li, (2 * GP,))
            alloc('wh%d' % li, (2 * WH_FLOATS,))
            alloc('whb%d' % li, (2 * WH_FLOATS,))      # same numbers, fragment order of dz . Wh^T (BPTT)
            if self.side_dim(li):
                alloc('we', (self.side_p, 2 * GP))
        alloc('pw', (2 * HP, self.ldp))
        alloc('pb', (self.ldp,))
        self._mlp_shapes = []
        if self.mlp:
            W, P = self.mlp, self.mlp_in_pitch
            self._mlp_shapes = [('mw1a', (P, W)), ('mw1b', (P, W)), ('mb1', (W,)), ('mw2', (W, W)), ('mb2', (W,)),
                                ('mw3', (W, W)), ('mb3', (W,))]
            for name, shape in self._mlp_shapes:
                alloc(name, shape)
        self.packed_size = poff

        # the two index maps depend on the constructor arguments alone and cost ~0.1 s of numpy to build: every model
        # of the same shape in a process (bench entries, the inference model next to the training one) shares them
        key = (self.input_dim, net_dim, F, self.side, mlp, mlp_in_pitch, self.asr)
        cached = _INDEX_CACHE.get(key)
        self.pack_index = cached[0] if cached else self._build_pack_index()     # packed pos -> ref pos (ref_size = zero slot)

        # ---- gradient layout produced by the backward kernels (GEMM outputs, natural k-major form)
        self.gpacked = {}
        goff = 0

        def galloc(name, shape):
            nonlocal goff
            self.gpacked[name] = (goff, shape)
            goff += round_up(int(np.prod(shape)), 64)
        for li in range(self.num_layers):
            galloc('dwx%d' % li, (self.kp[li], 2 * GP))
            galloc('db%d' % li, (2 * GP,))
            galloc('dwh%d' % li, (2, HP, GP))
            if self.side_dim(li):
                galloc('dwe', (self.side_p, 2 * GP))
        galloc('dpw', (2 * HP, self.ldp))
        galloc('dpb', (self.ldp,))
        for name, shape in self._mlp_shapes:
            galloc('d' + name, shape)
        self.gpacked_size = goff
        self.grad_index = cached[1] if cached else self._build_grad_index()     # ref pos -> gpacked pos
        if not cached:
            _INDEX_CACHE[key] = (self.pack_index, self.grad_index)

    def signature(self):
        """Shape signature stored in checkpoints."""
        sig = ([self.input_dim, self.H, self.num_layers, self.F] + (list(self.side) if self.side else [])
               + ([-self.mlp] if self.mlp else []) + ([-100000 - self.asr] if self.asr else []))
        if len(set(self.Hs)) > 1:                      # unequal widths: spelled out behind a marker (equal: round 1's form)
            sig += [-200000] + list(self.Hs)
        return sig

    def _mlp_map(self):
        """[(packed name, reference name, row offset in the reference matrix, rows)] of the dense MLP entries."""
        F, W = self.F, self.mlp
        return [('mw1a', 'weights_1', 0, F), ('mw1b', 'weights_1', F, F), ('mb1', 'biases_1', 0, 0),
                ('mw2', 'weights_2', 0, W), ('mb2', 'biases_2', 0, 0), ('mw3', 'weights_3', 0, W), ('mb3', 'biases_3', 0, 0)]

    def side_dim(self, li):
        return self.side[1] if self.side is not None and self.side[0] == li else 0

    # ------------------------------------------------------------------------------
    def input_row_map(self, li):
        """Padded input column k of layer li -> row of the reference kernel (or -1)."""
        H = self.Hs[li - 1] if li > 0 else 0           # units per direction of the layer below
        k = np.arange(self.kp[li])
        if li == 0:
            return np.where(k < self.input_dim, k, -1)
        r = np.full(self.kp[li], -1, dtype=np.int64)
        r[:H] = np.arange(H)
        r[HP:HP + H] = H + np.arange(H)
        return r

    def _build_pack_index(self):
        F, Z = self.F, self.ref_size
        idx = np.full(self.packed_size, Z, dtype=np.int64)
        # the two fragment orders of the recurrent kernel do not depend on direction or gate, and on the layer only through
        # its width: positions and (k, unit) of the cells that hold a weight, once per width
        #   forward  [d][w][q][g][lane][s]:  Wh[k = 8q + 4(lane >> 5) + s][unit = 32w + (lane & 31)] of gate g
        fw_, fq_, flane_, fs_ = np.meshgrid(np.arange(8), np.arange(32), np.arange(64), np.arange(4), indexing='ij')
        kk, uu = 8 * fq_ + 4 * (flane_ >> 5) + fs_, 32 * fw_ + (flane_ & 31)
        fposall = ((fw_ * 32 + fq_) * 4) * 256 + flane_ * 4 + fs_
        #   transposed product [d][w][q 128][lane][s]:  Wh[unit' = 32w + (lane & 31)][packed col = 8q + 4(lane >> 5) + s]
        w_, q_, lane_, s_ = np.meshgrid(np.arange(8), np.arange(128), np.arange(64), np.arange(4), indexing='ij')
        col = 8 * q_ + 4 * (lane_ >> 5) + s_              # packed col inside this direction
        up = 32 * w_ + (lane_ & 31)                       # unit' (row of Wh)
        cg = (col % 128) // 32                            # gate of that column
        cu = (col // 128) * 32 + col % 32                 # hidden unit of that column
        posb = ((w_ * 128 + q_) * 64 + lane_) * 4 + s_
        frag = {}

        def fragments(H):
            if H not in frag:
                ok = (kk < H) & (uu < H)
                okb = (up < H) & (cu < H)
                t_pos, t_up, t_cu = [], [], []
                for g in range(4):
                    sel = okb & (cg == g)
                    t_pos.append(posb[sel]), t_up.append(up[sel]), t_cu.append(cu[sel])
                frag[H] = (fposall[ok], kk[ok], uu[ok], t_pos, t_up, t_cu)
            return frag[H]
        for li in range(self.num_layers):
            H = self.Hs[li]
            u = np.arange(H)
            f_pos, f_k, f_u, t_pos, t_up, t_cu = fragments(H)
            E = self.side_dim(li)
            D = self.in_dims[li] + E                                    # first recurrent row of the TF kernel
            rmap = self.input_row_map(li)
            wx_off, (kp, _) = self.packed['wx%d' % li]
            b_off, _ = self.packed['b%d' % li]
            wh_off, _ = self.packed['wh%d' % li]
            for d, dname in enumerate(('fw', 'bw')):
                k_off = self._ref_off['cell_%d/%s/kernel' % (li, dname)]
                bias_off = self._ref_off['cell_%d/%s/bias' % (li, dname)]
                for g in range(4):
                    cols = packed_gate_col(d, g, u)                      # [H]
                    # input kernel rows
                    rows = np.nonzero(rmap >= 0)[0]
                    src = k_off + rmap[rows][:, None] * (4 * H) + (g * H + u)[None, :]
                    idx[wx_off + rows[:, None] * (2 * GP) + cols[None, :]] = src
                    idx[b_off + cols] = bias_off + g * H + u
                    if E:
                        we_off, _ = self.packed['we']
                        er = np.arange(E)
                        idx[we_off + er[:, None] * (2 * GP) + cols[None, :]] = (
                            k_off + (self.in_dims[li] + er)[:, None] * (4 * H) + (g * H + u)[None, :])
                    # recurrent kernel in fragment order [d][w][q][g][lane][s]
                    idx[wh_off + d * (8 * 32 * 4 * 256) + g * 256 + f_pos] = k_off + (D + f_k) * (4 * H) + g * H + f_u
                    # transposed-product fragment order [d][w][q 128][lane][s]
                    whb_off, _ = self.packed['whb%d' % li]
                    idx[whb_off + d * (8 * 128 * 256) + t_pos[g]] = k_off + (D + t_up[g]) * (4 * H) + g * H + t_cu[g]
        H = self.H                                       # the top layer feeds the projection
        pw_off, _ = self.packed['pw']
        pb_off, _ = self.packed['pb']
        rmap = np.full(2 * HP, -1, dtype=np.int64)
        rmap[:H] = np.arange(H)
        rmap[HP:HP + H] = H + np.arange(H)
        rows = np.nonzero(rmap >= 0)[0]
        c = np.arange(F)
        idx[pw_off + rows[:, None] * self.ldp + c[None, :]] = (
            self._ref_off['logits/weights'] + rmap[rows][:, None] * F + c[None, :])
        idx[pb_off + c] = self._ref_off['logits/biases'] + c
        if self.asr:
            a = np.arange(self.asr)
            idx[pw_off + rows[:, None] * self.ldp + (self.asr_col + a)[None, :]] = (
                self._ref_off['asr/weights'] + rmap[rows][:, None] * self.asr + a[None, :])
            idx[pb_off + self.asr_col + a] = self._ref_off['asr/biases'] + a
        if self.mlp:
            W = self.mlp
            w = np.arange(W)
            for pname, rname, r0, rows in self._mlp_map():
                poff, _ = self.packed[pname]
                roff = self._ref_off['speaker_embedding/' + rname]
                if rows:
                    r = np.arange(rows)
                    idx[poff + r[:, None] * W + w[None, :]] = roff + (r0 + r)[:, None] * W + w[None, :]
                else:
                    idx[poff + w] = roff + w
        return idx

    def _build_grad_index(self):
        """Position of every reference parameter's gradient inside the gpacked buffer."""
        F = self.F
        gi = np.full(self.ref_size, -1, dtype=np.int64)
        for li in range(self.num_layers):
            H = self.Hs[li]
            u = np.arange(H)
            E = self.side_dim(li)
            Dm = self.in_dims[li]
            D = Dm + E
            rmap = self.input_row_map(li)
            krows = np.nonzero(rmap >= 0)[0]                       # padded k for reference rows 0..Dm-1 (in order)
            assert (rmap[krows] == np.arange(Dm)).all()
            dwx_off, _ = self.gpacked['dwx%d' % li]
            db_off, _ = self.gpacked['db%d' % li]
            dwh_off, _ = self.gpacked['dwh%d' % li]
            for d, dname in enumerate(('fw', 'bw')):
                k_off = self._ref_off['cell_%d/%s/kernel' % (li, dname)]
                bias_off = self._ref_off['cell_%d/%s/bias' % (li, dname)]
                for g in range(4):
                    cols = packed_gate_col(d, g, u)
                    lcols = packed_gate_col(0, g, u)
                    gi[k_off + np.arange(Dm)[:, None] * (4 * H) + (g * H + u)[None, :]] = (
                        dwx_off + krows[:, None] * (2 * GP) + cols[None, :])
                    if E:
                        dwe_off, _ = self.gpacked['dwe']
                        gi[k_off + (Dm + np.arange(E))[:, None] * (4 * H) + (g * H + u)[None, :]] = (
                            dwe_off + np.arange(E)[:, None] * (2 * GP) + cols[None, :])
                    gi[k_off + (D + np.arange(H))[:, None] * (4 * H) + (g * H + u)[None, :]] = (
                        dwh_off + (d * HP + np.arange(H))[:, None] * GP + lcols[None, :])
                    if self.ones_col[li] >= 0:
                        gi[bias_off + g * H + u] = dwx_off + self.ones_col[li] * (2 * GP) + cols
                    else:
                        gi[bias_off + g * H + u] = db_off + cols
        H = self.H
        dpw_off, _ = self.gpacked['dpw']
        dpb_off, _ = self.gpacked['dpb']
        prow = np.concatenate([np.arange(H), HP + np.arange(H)])
        c = np.arange(F)
        gi[self._ref_off['logits/weights'] + np.arange(2 * H)[:, None] * F + c[None, :]] = (
            dpw_off + prow[:, None] * self.ldp + c[None, :])
        top = dpw_off + self.ones_col_top * self.ldp if self.ones_col_top >= 0 else dpb_off
        gi[self._ref_off['logits/biases'] + c] = top + c
        if self.asr:
            a = np.arange(self.asr)
            gi[self._ref_off['asr/weights'] + np.arange(2 * H)[:, None] * self.asr + a[None, :]] = (
                dpw_off + prow[:, None] * self.ldp + (self.asr_col + a)[None, :])
            gi[self._ref_off['asr/biases'] + a] = top + self.asr_col + a
        if self.mlp:
            W = self.mlp
            w = np.arange(W)
            for pname, rname, r0, rows in self._mlp_map():
                goff, _ = self.gpacked['d' + pname]
                roff = self._ref_off['speaker_embedding/' + rname]
                if rows:
                    r = np.arange(rows)
                    gi[roff + (r0 + r)[:, None] * W + w[None, :]] = goff + r[:, None] * W + w[None, :]
                else:
                    gi[roff + w] = goff + w
        assert (gi >= 0).all()
        return gi

    def gpacked_view(self, flat, name):
        off, shape = self.gpacked[name]
        return flat[off: off + int(np.prod(shape))].reshape(shape)

    # ------------------------------------------------------------------------------
    def ref_view(self, flat, name):
        """View of variable `name` inside a flat reference-layout buffer (numpy or torch)."""
        for n, shape, off in self.ref_entries:
            if n == name:
                return flat[off: off + int(np.prod(shape))].reshape(shape)
        raise KeyError(name)

    def packed_view(self, flat, name):
        off, shape = self.packed[name]
        return flat[off: off + int(np.prod(shape))].reshape(shape)

    def flatten_oracle_params(self, params):
        """oracle.blstm params dict -> flat reference-layout float32 numpy array."""
        flat = np.zeros(self.ref_size, dtype=np.float32)
        for li, layer in enumerate(params['layers']):
            for dname in ('fw', 'bw'):
                self.ref_view(flat, 'cell_%d/%s/kernel' % (li, dname))[...] = layer[dname]['kernel']
                self.ref_view(flat, 'cell_%d/%s/bias' % (li, dname))[...] = layer[dname]['bias']
        self.ref_view(flat, 'logits/weights')[...] = params['proj']['weights']
        self.ref_view(flat, 'logits/biases')[...] = params['proj']['biases']
        if 'asr' in params:
            self.ref_view(flat, 'asr/weights')[...] = params['asr']['weights']
            self.ref_view(flat, 'asr/biases')[...] = params['asr']['biases']
        for k, v in params.get('mlp', {}).items():
            self.ref_view(flat, 'speaker_embedding/' + k)[...] = v
        return flat

    def unflatten_to_oracle_params(self, flat):
        flat = np.asarray(flat)
        return {
            'layers': [{dname: {'kernel': np.array(self.ref_view(flat, 'cell_%d/%s/kernel' % (li, dname))),
                                'bias': np.array(self.ref_view(flat, 'cell_%d/%s/bias' % (li, dname)))}
                        for dname in ('fw', 'bw')} for li in range(self.num_layers)],
            'proj': {'weights': np.array(self.ref_view(flat, 'logits/weights')),
                     'biases': np.array(self.ref_view(flat, 'logits/biases'))},
        }
