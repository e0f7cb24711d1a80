"""EGM warm start of BGM -- INTERIM implementation.

Mirrors /root/reference/src/bayesgm/models/bgm/base.py:
    train_disc_step :190-244   train_gen_step :246-289   egm_init :292-340
and the Discriminator of models/networks/base.py:338-385 (Dense -> BatchNorm(batch statistics) -> tanh).

CausalBGM's warm start is native (csrc/egm_kernels.h behind bgm_causal_egm_*).  BGM's still runs on the GPU
through PyTorch autograd: two discriminators (latent and data space, LSGAN targets 0.9 / 0.1, optional gradient
penalty) against a generator that is itself batch-normalised in training mode.  It is B=32 latency-bound work
outside SURVEY section 8's a-e rows; both steps are captured into HIP graphs so that an iteration costs two graph
launches.  The generator it trains is handed to the HIP engine afterwards.
"""
import numpy as np
import torch

LEAK = 0.2
B1, B2, ADAM_EPS = 0.9, 0.99, 1e-7   # causalbgm/base.py:86-87
BN_EPS = 1e-3                        # keras BatchNormalization default


def _mlp(net, x):
    h = x
    for i, (W, b) in enumerate(net):
        h = h @ W + b
        if i < len(net) - 1:
            h = torch.maximum(h, LEAK * h)
    return h


def _disc(dnet, x):
    """Discriminator.call (networks/base.py:364-385): BatchNorm in training mode (batch statistics)."""
    h = x
    n = len(dnet["W"])
    for i in range(n - 1):
        h = h @ dnet["W"][i] + dnet["b"][i]
        mu = h.mean(dim=0)
        var = h.var(dim=0, unbiased=False)
        h = (h - mu) / torch.sqrt(var + BN_EPS) * dnet["gamma"][i] + dnet["beta"][i]
        h = torch.tanh(h)
    return h @ dnet["W"][n - 1] + dnet["b"][n - 1]


class _KerasAdam(object):
    """tf.keras.optimizers.Adam (optimizer_v2) update written with torch ops (graph-capturable)."""

    def __init__(self, params, lr, b1=B1, b2=B2):
        self.params = params
        self.lr = float(lr)
        self.b1, self.b2 = float(b1), float(b2)
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = torch.zeros((), device=params[0].device, dtype=torch.float64)

    def step(self, grads):
        self.t += 1
        b1, b2 = self.b1, self.b2
        lr_t = (self.lr * torch.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)).float()
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            p.sub_(lr_t * m / (torch.sqrt(v) + ADAM_EPS))


def _glorot(rs, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


# =====================================================================================================
# BGM
# =====================================================================================================
BN_MOMENTUM = 0.99   # keras BatchNormalization default


def _new_disc(rs, in_dim, units, device):
    dims = [in_dim] + list(units) + [1]
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=device)
    return {"W": [t(_glorot(rs, dims[i], dims[i + 1])) for i in range(len(dims) - 1)],
            "b": [torch.zeros(dims[i + 1], device=device) for i in range(len(dims) - 1)],
            "gamma": [torch.ones(dims[i + 1], device=device) for i in range(len(dims) - 2)],
            "beta": [torch.zeros(dims[i + 1], device=device) for i in range(len(dims) - 2)]}


class BgmEGM(object):
    """EGM warm start of BGM: generator g (BaseVariationalNet, called with training=True: the input
    BatchNorm uses batch statistics and every call moves its moving averages), encoder e, LSGAN
    discriminators dz, dx (targets 0.9 / 0.1) with optional gradient penalty `gamma`."""

    def __init__(self, g, params, device, rs, batch_size=32):
        self.p = params
        self.dev = device
        self.B = batch_size
        t = lambda a: torch.tensor(np.asarray(a, np.float32), device=device)
        q, xd = int(params["z_dim"]), int(params["x_dim"])
        self.q, self.xd = q, xd
        self.g = {"gamma": t(g["bn"]["gamma"]), "beta": t(g["bn"]["beta"]),
                  "mmean": t(g["bn"]["mean"]), "mvar": t(g["bn"]["var"]),
                  "trunk": [(t(W), t(b)) for W, b in g["trunk"]],
                  "mean": (t(g["mean"][0]), t(g["mean"][1])), "var": (t(g["var"][0]), t(g["var"][1]))}
        dims = [xd] + list(params["e_units"]) + [q]
        self.e = [(t(_glorot(rs, dims[i], dims[i + 1])), torch.zeros(dims[i + 1], device=device)) for i in range(len(dims) - 1)]
        self.dz = _new_disc(rs, q, params["dz_units"], device)
        self.dx = _new_disc(rs, xd, params["dx_units"], device)
        self.gen_params = ([self.g["gamma"], self.g["beta"]] + [a for Wb in self.g["trunk"] for a in Wb] +
                           list(self.g["mean"]) + list(self.g["var"]) + [a for Wb in self.e for a in Wb])
        self.disc_params = (self.dz["W"] + self.dz["b"] + self.dz["gamma"] + self.dz["beta"] +
                            self.dx["W"] + self.dx["b"] + self.dx["gamma"] + self.dx["beta"])
        for a in self.gen_params + self.disc_params:
            a.requires_grad_(True)
        self.g_opt = _KerasAdam(self.gen_params, params["lr"], 0.5, 0.9)    # bgm/base.py:82-85
        self.d_opt = _KerasAdam(self.disc_params, params["lr"], 0.5, 0.9)
        self.in_z = torch.zeros(batch_size, q, device=device)
        self.in_x = torch.zeros(batch_size, xd, device=device)
        self.in_n1 = torch.zeros(batch_size, xd, device=device)
        self.in_n2 = torch.zeros(batch_size, xd, device=device)
        self.in_eps = torch.zeros(2, device=device)
        self.out_d = torch.zeros(3, device=device)
        self.out_g = torch.zeros(6, device=device)
        self._graph_d = self._graph_g = None

    # ---- generator in training mode (networks/base.py:98-111) ---------------------------------------
    def _g_train(self, z):
        g = self.g
        mu = z.mean(dim=0)
        var = z.var(dim=0, unbiased=False)
        with torch.no_grad():     # moving = moving * momentum + batch * (1 - momentum)
            g["mmean"].mul_(BN_MOMENTUM).add_(mu.detach(), alpha=1 - BN_MOMENTUM)
            g["mvar"].mul_(BN_MOMENTUM).add_(var.detach(), alpha=1 - BN_MOMENTUM)
        h = (z - mu) / torch.sqrt(var + BN_EPS) * g["gamma"] + g["beta"]
        return self._g_tail(h)

    def _g_tail(self, h):
        g = self.g
        for W, b in g["trunk"]:
            h = h @ W + b
            h = torch.maximum(h, LEAK * h)
        mean = h @ g["mean"][0] + g["mean"][1]
        var = torch.nn.functional.softplus(h @ g["var"][0] + g["var"][1]) + 1e-6
        return mean, var

    def g_infer(self, z):
        """g_net(z, training=False): moving statistics."""
        g = self.g
        h = (z - g["mmean"]) / torch.sqrt(g["mvar"] + BN_EPS) * g["gamma"] + g["beta"]
        return self._g_tail(h)

    def encode(self, x):
        with torch.no_grad():
            return _mlp(self.e, x)

    def eval_pass(self, data):
        """The evaluation block of egm_init (bgm/base.py:312-317): z = e(data); g_net(z) is called with its
        default training=True, so the BatchNorm moving statistics move towards those of e(data)."""
        with torch.no_grad():
            z = _mlp(self.e, data)
            x_rec, _ = self._g_train(z)
            return z, x_rec, float(((data - x_rec) ** 2).mean().item())

    # ---- the two step functions ---------------------------------------------------------------------
    def _disc_step(self):
        z, x = self.in_z, self.in_x
        eps_z, eps_x = self.in_eps[0], self.in_eps[1]
        gam = float(self.p["gamma"])
        with torch.no_grad():
            z_ = _mlp(self.e, x)
            mu_, s2_ = self._g_train(z)
            x_ = self.in_n1 * torch.sqrt(s2_) + mu_
        dx_, dz_ = _disc(self.dx, x_), _disc(self.dz, z_)
        dx, dz = _disc(self.dx, x), _disc(self.dz, z)
        dz_loss = (((0.9 - dz) ** 2).mean() + ((0.1 - dz_) ** 2).mean()) / 2.0
        dx_loss = (((0.9 - dx) ** 2).mean() + ((0.1 - dx_) ** 2).mean()) / 2.0
        d_loss = dx_loss + dz_loss
        if gam != 0.0:
            z_hat = (z * eps_z + z_ * (1 - eps_z)).requires_grad_(True)
            x_hat = (x * eps_x + x_ * (1 - eps_x)).requires_grad_(True)
            (gz,) = torch.autograd.grad(_disc(self.dz, z_hat).sum(), z_hat, create_graph=True)
            (gx,) = torch.autograd.grad(_disc(self.dx, x_hat).sum(), x_hat, create_graph=True)
            gpz = ((torch.sqrt((gz ** 2).sum(dim=1)) - 1.0) ** 2).mean()
            gpx = ((torch.sqrt((gx ** 2).sum(dim=1)) - 1.0) ** 2).mean()
            d_loss = d_loss + gam * (gpz + gpx)
        grads = torch.autograd.grad(d_loss, self.disc_params, allow_unused=True)
        grads = [torch.zeros_like(p_) if g_ is None else g_ for g_, p_ in zip(grads, self.disc_params)]
        with torch.no_grad():
            self.d_opt.step(grads)
            self.out_d.copy_(torch.stack([dz_loss, dx_loss, d_loss]).detach())

    def _gen_step(self):
        z, x = self.in_z, self.in_x
        mu_, s2_ = self._g_train(z)
        x_ = self.in_n1 * torch.sqrt(s2_) + mu_
        reg = (s2_ ** 2).mean()
        z_ = _mlp(self.e, x)
        z__ = _mlp(self.e, x_)
        mu__, s2__ = self._g_train(z_)
        x__ = self.in_n2 * torch.sqrt(s2__) + mu__
        dx_, dz_ = _disc(self.dx, x_), _disc(self.dz, z_)
        l2_x = ((x - x__) ** 2).mean()
        l2_z = ((z - z__) ** 2).mean()
        g_adv = ((0.9 - dx_) ** 2).mean()
        e_adv = ((0.9 - dz_) ** 2).mean()
        loss = g_adv + e_adv + 10 * (l2_x + l2_z) + float(self.p["alpha"]) * reg
        grads = torch.autograd.grad(loss, self.gen_params)
        with torch.no_grad():
            self.g_opt.step(grads)
            self.out_g.copy_(torch.stack([g_adv, e_adv, l2_z, l2_x, reg, loss]).detach())

    def _snapshot(self):
        st = [a.detach().clone() for a in self.gen_params + self.disc_params + [self.g["mmean"], self.g["mvar"]]]
        op = [[t_.clone() for t_ in o.m + o.v] + [o.t.clone()] for o in (self.g_opt, self.d_opt)]
        return st, op

    def _restore(self, snap):
        st, op = snap
        with torch.no_grad():
            for a, b in zip(self.gen_params + self.disc_params + [self.g["mmean"], self.g["mvar"]], st):
                a.copy_(b)
            for o, sv in zip((self.g_opt, self.d_opt), op):
                for t_, b in zip(o.m + o.v, sv[:-1]):
                    t_.copy_(b)
                o.t.copy_(sv[-1])

    def capture(self):
        """Capture both steps into HIP graphs; the eager warm-up iterations run on state that is restored."""
        snap = self._snapshot()
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for _ in range(3):
                self._disc_step()
                self._gen_step()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        gd, gg = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gd):
            self._disc_step()
        with torch.cuda.graph(gg):
            self._gen_step()
        self._restore(snap)
        self._graph_d, self._graph_g = gd, gg

    def disc_step(self, z, x, eps_z, eps_x):
        self.in_z.copy_(z); self.in_x.copy_(x)
        self.in_eps[0] = float(eps_z); self.in_eps[1] = float(eps_x)
        self.in_n1.normal_()
        self._graph_d.replay() if self._graph_d is not None else self._disc_step()

    def gen_step(self, z, x):
        self.in_z.copy_(z); self.in_x.copy_(x)
        self.in_n1.normal_(); self.in_n2.normal_()
        self._graph_g.replay() if self._graph_g is not None else self._gen_step()

    def export_g(self):
        c = lambda a: a.detach().cpu().numpy().copy()
        g = self.g
        return {"bn": {"gamma": c(g["gamma"]), "beta": c(g["beta"]), "mean": c(g["mmean"]), "var": c(g["mvar"])},
                "trunk": [(c(W), c(b)) for W, b in g["trunk"]],
                "mean": (c(g["mean"][0]), c(g["mean"][1])), "var": (c(g["var"][0]), c(g["var"][1]))}
