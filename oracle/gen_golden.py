#!/usr/bin/env python3
"""Record golden vectors from the reference itself (build container only).

TEST INFRASTRUCTURE ONLY.  Imports /root/reference through oracle/ref_harness.py, drives its
classes with scripted / seeded inputs and writes small .npz fixtures into tests/golden/.
Only data (inputs + the reference's outputs) is written; no reference source text is stored.

  python oracle/gen_golden.py [traces] [views] [her] [window_her] [ppo] [predictor] [predictor_update] [occlusion] [mgstep] [soa] [pretrain]     (default: all)

The random draws of Twoarmy (np.random.choice calls in twoarmy_v{4,6}.py) are replaced by the
engine's counter-based Philox words (oracle/philox.py) through ref_harness.patched_choice, so
reference, oracle and HIP engine all consume identical draws per (env, step, slot).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import philox  # noqa: E402
import ref_harness as rh  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
SEED = 9981  # reference default, soa/train_ppo.py:25

ERR_CODE = {AttributeError: 1, AssertionError: 2, TypeError: 3}
OP_RESET = -1


# ----------------------------------------------------------------------------- draws
class PhiloxSlots:
    """draw_fn for ref_harness.SlotRecorder: identifies the slot from (lo, n) + call order in the step."""

    def __init__(self, seed, env_id):
        self.seed, self.env_id = seed, env_id
        self.t = 0            # index of the current step() call
        self.calls = []       # (lo, n) seen in the current step
        self.log = []         # (t, slot, lo, n, value)

    def begin_step(self, t):
        self.t, self.calls = t, []

    def slot_of(self, lo, n):
        if (lo, n) == (0, 10):
            return philox.S_GATE
        if (lo, n) == (9, 4):
            return philox.S_WALL1
        if (lo, n) == (6, 4):
            return philox.S_WALL2 if (self.calls and self.calls[-1] == (9, 4)) else philox.S_SPAWN
        if (lo, n) == (4, 1):
            return 6          # choice(range(4,5)): single outcome
        if (lo, n) == (0, 2):
            return philox.S_COIN_B if (0, 2) in self.calls else philox.S_COIN_A
        raise AssertionError("unexpected draw range (%d,%d)" % (lo, n))

    def __call__(self, lo, n):
        slot = self.slot_of(lo, n)
        self.calls.append((lo, n))
        w = int(philox.draw_word(self.seed, self.env_id, self.t, slot))
        v = lo + w % n
        self.log.append((self.t, slot, lo, n, v))
        return v


def snapshot(env):
    def pos(objs):
        return [(-1, -1) if o.cur_pos is None else tuple(int(v) for v in o.cur_pos) for o in objs]
    return dict(
        agent=np.array(env.agent_pos, np.int32),
        scal=np.array([env.step_count, env.step_move, env.pone, env.patrol, env.up1, env.right2,
                       env.Update_longitudinal, env.Update_horizontal, env.risk_count,
                       env.first_to_room2, env.agent_dir], np.int32),
        balls=np.array(pos(env.obstacles), np.int32),
        o1=np.array(pos(env.obstacles1), np.int32),
        o2=np.array(pos(env.obstacles2), np.int32),
        grid=env.grid.encode().copy(),
    )


SCAL_NAMES = ["step_count", "step_move", "pone", "patrol", "up1", "right2", "Update_longitudinal",
              "Update_horizontal", "risk_count", "first_to_room2", "agent_dir"]


def run_trace(variant, ops, env_id, seed=SEED, natural_rng_seed=None):
    """Replay `ops` (env actions >= 0, OP_RESET) on a fresh reference env; returns dict of arrays."""
    env_buffer, _ = rh.soa_modules()
    et = env_buffer.Env_transact()
    slots = PhiloxSlots(seed, env_id)
    nat_log = []
    if natural_rng_seed is None:
        rec = rh.SlotRecorder(slots)
    else:
        # K8: the reference's own MT19937 stream, recorded so it can be replayed as explicit draws
        np.random.seed(natural_rng_seed)
        real_choice = np.random.choice

        def nat(lo, n):
            v = int(real_choice(range(lo, lo + n), 1).item()) if n > 1 else lo
            slot = slots.slot_of(lo, n)
            slots.calls.append((lo, n))
            nat_log.append((slots.t, slot, lo, n, v))
            return v
        rec = rh.SlotRecorder(nat)
    with rh.patched_choice(rec):
        env = rh.make_env(variant)
        rows = []
        t = 0
        for op in ops:
            row = dict(op=op, err=0, obs=np.zeros((17, 17, 3), np.uint8), reward=np.nan, term=0, trunc=0)
            if op == OP_RESET:
                row["obs"] = env.reset()["image"].copy()
            else:
                slots.begin_step(t)
                t += 1
                try:
                    obs, r, te, tr, _ = env.step(op)
                    row.update(obs=obs["image"].copy(), reward=float(r), term=int(te), trunc=int(tr))
                except tuple(ERR_CODE) as ex:
                    row["err"] = ERR_CODE[type(ex)]
            row.update(snapshot(env))
            row["matrix"] = et.matrix_env(env).copy()
            a, g = et.data_env(env)
            row["pos"] = np.concatenate([a, g])
            rows.append(row)
    out = {k: np.stack([np.asarray(r[k]) for r in rows]) for k in rows[0]}
    log = nat_log if natural_rng_seed is not None else slots.log
    out["draw_log"] = np.array(log, np.int64).reshape(-1, 5)
    out["variant"] = np.int32(4 if variant == "v4" else 6)
    out["env_id"] = np.int32(env_id)
    out["natural"] = np.int32(natural_rng_seed is not None)
    return out


def random_ops(rs, n, p=(0.2, 0.2, 0.2, 0.2, 0.2), reset_after_done=True):
    """Policy indices -> env actions (4 -> 6).  Resets are inserted lazily by the caller."""
    idx = rs.choice(5, size=n, p=p)
    return [6 if a == 4 else int(a) for a in idx]


def gen_traces():
    traces = []
    R = OP_RESET
    # --- scripted v6 known-answer traces
    v6_scripts = {
        "K1_wall_drop": [1, 6, 6],
        "K2_blocked_room2": [1, 1, 1] + [2] * 12,
        "K3_risk_trunc": [1] * 4 + [2] * 6 + [6] * 20,
        "K4_goal": [1] * 7 + [2] * 7 + [6, 6] + [2] * 6 + [1] * 4 + [6, 2, 1, R, 1, 2],
        "K5_ball_onto_agent": [1] * 7 + [2] * 7 + [6] * 5 + [6, 6, 2, 2, R, 2],
        "K6_timeout": [0] * 50 + [0, 1, R, 1],
        "K7_illegal": [4, 5, 7, 99, 1, 4, 2, 5],
        "K9_midreset": [1, 1, 2, 2, R, 2, 2, 1, 1, R, 1] + [2] * 10,
        "K10_nodone_reset": [1] * 7 + [2] * 7 + [6] * 5 + [6] * 40 + [1] * 30,  # keep stepping past done, never reset
        "K11_drift": [1] * 4 + [2] * 6 + [6] * 20 + [6] * 200,                 # risk truncations without reset: ball drift
    }
    eid = 0
    for name, ops in v6_scripts.items():
        tr = run_trace("v6", ops, eid)
        tr["name"] = np.array(name)
        traces.append(tr)
        eid += 1
    # --- random v6 traces with the training loop's reset-after-done
    for k in range(6):
        rs = np.random.RandomState(1000 + k)
        p = (0.2,) * 5 if k < 3 else (0.1, 0.3, 0.35, 0.1, 0.15)
        traces.append(_random_trace("v6", rs, 160, p, eid, "rand_v6_%d" % k))
        eid += 1
    # --- v4: scripted path into room 2 (K8 path) then wander, Philox draws
    path = [1] * 7 + [2] * 7
    v4_scripts = {
        "v4_room2_wait": path + [2] + [6] * 40,
        "v4_room2_goal": path + [2] * 7 + [1] * 4 + [6, 6],
        "v4_room2_patrol_left": path + [2, 2, 2] + [0] * 6 + [2, 2, 1, 1] + [6] * 10,
        "v4_room2_patrol_right": path + [2, 2] + [1] * 2 + [2] * 3 + [6] * 12,
        "v4_illegal_midreset": path + [2, 2, 4, 2, R, 2, 2],   # reset with patrol=True: TypeError on next step
        "v4_nodone": path + [2] * 3 + [6] * 120,
    }
    for name, ops in v4_scripts.items():
        for rep in range(2):   # two env ids -> different draws
            tr = run_trace("v4", ops, eid)
            tr["name"] = np.array("%s_%d" % (name, rep))
            traces.append(tr)
            eid += 1
    for k in range(10):
        rs = np.random.RandomState(2000 + k)
        p = (0.1, 0.3, 0.35, 0.1, 0.15) if k % 2 == 0 else (0.15, 0.25, 0.3, 0.1, 0.2)
        traces.append(_random_trace("v4", rs, 200, p, eid, "rand_v4_%d" % k))
        eid += 1
    # --- K8: the reference's natural MT19937 stream, np.random.seed(9981)
    tr = run_trace("v4", path + [2, 2, 2] + [6] * 30, eid, natural_rng_seed=9981)
    tr["name"] = np.array("K8_natural_seed9981")
    traces.append(tr)
    eid += 1
    tr = run_trace("v6", [1] * 7 + [2] * 7 + [6] * 5 + [R] + [1] * 4 + [2] * 6 + [6] * 20, eid, natural_rng_seed=7)
    tr["name"] = np.array("v6_natural_seed7")
    traces.append(tr)
    eid += 1

    flat = {}
    for i, tr in enumerate(traces):
        for k, v in tr.items():
            if k in ("grid", "obs"):
                v = v.astype(np.uint8)
            flat["t%02d_%s" % (i, k)] = v
    flat["n_traces"] = np.int32(len(traces))
    flat["scal_names"] = np.array(SCAL_NAMES)
    flat["seed"] = np.int64(SEED)
    path_out = os.path.join(GOLD, "twoarmy_traces.npz")
    np.savez_compressed(path_out, **flat)
    nsteps = sum(len(tr["op"]) for tr in traces)
    print("traces: %d traces, %d ops -> %s (%.1f KB)" % (len(traces), nsteps, path_out,
                                                         os.path.getsize(path_out) / 1024))


def _random_trace(variant, rs, n, p, eid, name):
    """Random policy with reset after every done step (soa/train_ppo.py:104,126,154)."""
    # Two-pass: first discover where dones happen (needs the env), so generate ops online.
    ops = []
    env_actions = random_ops(rs, n, p)
    # online replay to insert resets
    slots = PhiloxSlots(SEED, eid)
    rec = rh.SlotRecorder(slots)
    with rh.patched_choice(rec):
        env = rh.make_env(variant)
        t = 0
        for a in env_actions:
            slots.begin_step(t)
            t += 1
            _, _, te, tr, _ = env.step(a)
            ops.append(a)
            if te or tr:
                env.reset()
                ops.append(OP_RESET)
    tr = run_trace(variant, ops, eid)
    tr["name"] = np.array(name)
    return tr


# ----------------------------------------------------------------------------- views
def gen_views():
    """gen_obs_grid(V).encode() for dirs 0-3 x V in {3,5,7,17} on rich v4 states (minigrid.py:1443-1478)."""
    slots = PhiloxSlots(SEED, 900)
    rec = rh.SlotRecorder(slots)
    grids, agents, dirs, views, images = [], [], [], [], {}
    with rh.patched_choice(rec):
        env = rh.make_env("v4")
        ops = [1] * 7 + [2] * 7 + [2, 2, 6, 6]
        for t, a in enumerate(ops):
            slots.begin_step(t)
            env.step(a)
        positions = [(1, 1), (15, 15), (1, 15), (15, 1), (8, 8), (3, 15), (10, 6), (13, 3), (6, 10), (14, 2), (5, 12)]
        case = 0
        for (x, y) in positions:
            for d in range(4):
                env.agent_pos = (x, y)
                env.agent_dir = d
                for V in (3, 5, 7, 17):
                    g, _ = env.gen_obs_grid(V)
                    images["img_%03d_V%d" % (case, V)] = g.encode().astype(np.uint8)
                grids.append(env.grid.encode().astype(np.uint8))
                agents.append((x, y))
                dirs.append(d)
                case += 1
    out = dict(grid=np.stack(grids), agent=np.array(agents, np.int32), dir=np.array(dirs, np.int32),
               view_sizes=np.array([3, 5, 7, 17], np.int32), **images)
    path_out = os.path.join(GOLD, "views.npz")
    np.savez_compressed(path_out, **out)
    print("views: %d cases x 4 sizes -> %s (%.1f KB)" % (len(grids), path_out, os.path.getsize(path_out) / 1024))


# ----------------------------------------------------------------------------- PPO
def det_weights(module, seed):
    """Deterministic, torch-RNG-free weights: w[i] = scale * sin(0.37 * i + k + seed), scale ~ xavier."""
    import torch
    sd = {}
    for k, (name, prm) in enumerate(module.state_dict().items()):
        n = prm.numel()
        fan = max(1, n // prm.shape[0]) if prm.dim() > 1 else 1
        scale = (1.5 / np.sqrt(fan)) if prm.dim() > 1 else 0.05
        v = scale * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.7 * k + seed)
        sd[name] = torch.tensor(v.reshape(tuple(prm.shape)), dtype=prm.dtype)
    return sd


def param_stats(module):
    out = []
    for name, prm in module.state_dict().items():
        v = prm.detach().double().reshape(-1)
        out.append((name, tuple(prm.shape), float(v.sum()), float(v.abs().sum()), v[:4].tolist()))
    return out


def collect_buffer(env_buffer, variant, n_records, seed):
    """Fill a Buffer_gridworld exactly like soa/train_ppo.py:93-123 with a seeded random policy (HER off)."""
    buf = env_buffer.Buffer_gridworld()
    buf.grid_size = 17
    buf.transition = np.dtype([('s', np.float32, (5, 289)), ('a', np.int64, (1,)), ('p', np.float32, (5, 2)),
                               ('g', np.float32, (2,)), ('r', np.float32, (1,)), ('d', np.float32, (1,)),
                               ('a_logp', np.float32, (1,))])
    buf.buffer_capacity = n_records
    buf.buffer = np.empty(n_records, dtype=buf.transition)
    rs = np.random.RandomState(seed)
    slots = PhiloxSlots(SEED, 700 + seed)
    rec = rh.SlotRecorder(slots)

    class Args:
        server = True

    class Win:
        def set_caption(self, *_):
            pass

        def show_img(self, *_):
            pass
    t = 0
    buf.episode_ends = []                   # index of the last record of every finished episode (harness bookkeeping)
    with rh.patched_choice(rec):
        env = rh.make_env(variant)
        while not buf.full:
            et = env_buffer.Env_transact()
            sm_stack, st_stack, goal = et.reset(env, Win())
            for _ in range(10000):
                a_idx = int(rs.choice(5, p=(0.1, 0.3, 0.35, 0.1, 0.15)))
                logp = float(np.log(0.2) - 0.01 * rs.rand())
                action = et.env_action(env, a_idx)
                slots.begin_step(t)
                t += 1
                _, r, term, trunc, done = et.step(env, None, action, Args)
                state, goal = et.data_env(env)
                st_stack = np.append(np.delete(st_stack, 0, 0), [state], 0)
                sm_stack = np.append(np.delete(sm_stack, 0, 0), [et.matrix_env(env)], 0)
                buf.store((np.array(sm_stack, dtype='float32'), np.array([a_idx], dtype='int64'),
                           np.array(st_stack, dtype='float32'), np.array(goal, dtype='float32'),
                           np.array([r], dtype='float32'), np.array([done], dtype='int64'),
                           np.array([logp], dtype='float32')))
                if term or trunc:
                    buf.episode_ends.append((buf.counter - 1) % n_records)
                if term or trunc or buf.full:
                    break
    return buf


def gen_ppo():
    """soa/agent/PPO.py + soa/agent/net/all_net.py: init statistics, forward, log-prob/entropy, update losses."""
    import torch
    env_buffer, ppo_mod = rh.soa_modules()
    from agent.net.all_net import Net_PPO_actor, Net_PPO_critic
    out = {}
    # (1) init parity: same construction order as PPO.__init__ (PPO.py:46-47) under torch.manual_seed
    torch.manual_seed(SEED)
    actor, critic = Net_PPO_actor(), Net_PPO_critic()
    for tag, net in (("actor", actor), ("critic", critic)):
        st = param_stats(net)
        out["init_%s_names" % tag] = np.array([x[0] for x in st])
        out["init_%s_sum" % tag] = np.array([x[2] for x in st])
        out["init_%s_abs" % tag] = np.array([x[3] for x in st])
        out["init_%s_head" % tag] = np.array([x[4] + [0.0] * (4 - len(x[4])) for x in st])
        out["init_%s_numel" % tag] = np.array([int(np.prod(x[1])) for x in st])
    # (2) forward / distribution parity with injected deterministic weights
    buf = collect_buffer(env_buffer, "v6", 64, seed=3)
    b = buf.buffer
    actor.load_state_dict(det_weights(actor, 1))
    critic.load_state_dict(det_weights(critic, 2))
    actor.eval(); critic.eval()
    s = torch.tensor(b['s'][:12]); pp = torch.tensor(b['p'][:12]); g = torch.tensor(b['g'][:12])
    with torch.no_grad():
        probs = actor(s[:, 1:5], pp[:, 1:5], g)
        val = critic(s[:, 1:5], pp[:, 1:5], g)
        dist = torch.distributions.Categorical(probs=probs)
        a = torch.tensor(b['a'][:12, 0])
        out["fwd_probs"] = probs.numpy(); out["fwd_value"] = val.numpy()
        out["fwd_logp"] = dist.log_prob(a).numpy(); out["fwd_entropy"] = dist.entropy().numpy()
        out["fwd_logits_all"] = dist.logits.numpy()
    # (3) PPO.update: 64 records, batch 16, K_epochs 2 (PPO.py:103-161)
    agent = ppo_mod.PPO()
    agent.actor.load_state_dict(det_weights(agent.actor, 1))
    agent.critic.load_state_dict(det_weights(agent.critic, 2))
    agent.batch_size = 16
    agent.K_epochs = 2
    agent.heatmapfilename = "x"
    torch.manual_seed(123)
    perm_state = torch.get_rng_state()
    agent.update(b, torch.device("cpu"), 0)
    out["upd_action_loss"] = np.array([v for _, v in agent.writer.scalars["loss/action_loss_update"]])
    out["upd_value_loss"] = np.array([v for _, v in agent.writer.scalars["loss/value_loss_update"]])
    for tag, net in (("actor", agent.actor), ("critic", agent.critic)):
        st = param_stats(net)
        out["upd_%s_sum" % tag] = np.array([x[2] for x in st])
        out["upd_%s_abs" % tag] = np.array([x[3] for x in st])
    # the minibatch permutations torch drew (SubsetRandomSampler == randperm per epoch)
    torch.set_rng_state(perm_state)
    out["upd_perms"] = np.stack([torch.randperm(64).numpy() for _ in range(2)])
    for k in ('s', 'a', 'p', 'g', 'r', 'd', 'a_logp'):
        out["buf_" + k] = b[k]
    out["hyper"] = np.array([agent.gamma, agent.clip_param, agent.entropy_coef, 1e-4, 1e-5])
    path_out = os.path.join(GOLD, "ppo.npz")
    np.savez_compressed(path_out, **out)
    print("ppo: losses", out["upd_action_loss"][:3], out["upd_value_loss"][:3], "-> %s (%.1f KB)"
          % (path_out, os.path.getsize(path_out) / 1024))


# ----------------------------------------------------------------------------- HER
def gen_her():
    """Buffer_gridworld.her_func (soa/env_buffer.py:101-143) on recorded episodes, incl. ring wrap."""
    env_buffer, _ = rh.soa_modules()
    out = {}
    cases = [dict(cap=256, seed=0, pre=0, ep=0, variant="v6"), dict(cap=256, seed=1, pre=17, ep=1, variant="v6"),
             dict(cap=70, seed=2, pre=25, ep=0, variant="v6"), dict(cap=96, seed=5, pre=30, ep=2, variant="v4"),
             dict(cap=128, seed=7, pre=3, ep=1, variant="v4"), dict(cap=60, seed=11, pre=8, ep=0, variant="v4"),
             dict(cap=200, seed=13, pre=100, ep=3, variant="v6")]
    for ci, c in enumerate(cases):
        buf = collect_buffer(env_buffer, c["variant"], 256, seed=10 + ci)     # source of realistic records
        src = buf.buffer
        ends = buf.episode_ends
        first = 0 if c["ep"] == 0 else ends[c["ep"] - 1] + 1
        L = ends[c["ep"]] - first + 1
        ep = src[first:first + L].copy()
        b2 = env_buffer.Buffer_gridworld()
        b2.grid_size = 17
        b2.transition = src.dtype
        b2.buffer_capacity = c["cap"]
        b2.buffer = np.zeros(c["cap"], dtype=src.dtype)
        b2.counter = c["pre"]
        b2.epo_counter_start = c["pre"]
        for r in ep:
            b2.store(r)
        before = b2.buffer.copy()
        cnt_before, full_before = b2.counter, b2.full
        np.random.seed(c["seed"])
        b2.her_func(max_steps=50, newgoal_size_in=4)
        for k in src.dtype.names:
            out["c%d_before_%s" % (ci, k)] = before[k]
            out["c%d_after_%s" % (ci, k)] = b2.buffer[k]
        out["c%d_meta" % ci] = np.array([c["cap"], c["seed"], c["pre"], L, cnt_before, int(full_before),
                                         b2.counter, int(b2.full), b2.epo_counter_end])
    out["n_cases"] = np.int32(len(cases))
    path_out = os.path.join(GOLD, "her.npz")
    np.savez_compressed(path_out, **out)
    print("her: %d cases -> %s (%.1f KB)" % (len(cases), path_out, os.path.getsize(path_out) / 1024))


# ----------------------------------------------------------------------------- predictor path
# ----------------------------------------------------------------------------- HER on 9-frame window records
def collect_window_episode(env_buffer, variant, buf, seed, with_f):
    """One episode stored the way soa/train_ppo_predictor.py:105-171 (with_f: soa/train_SoA.py:130-183) stores it:
    9-frame windows delayed by four steps + the four terminal windows, through the reference's own pre_store()."""
    rs = np.random.RandomState(seed)
    slots = PhiloxSlots(SEED, 900 + seed)
    rec = rh.SlotRecorder(slots)

    class Args:
        server = True

    class Win:
        def set_caption(self, *_):
            pass

        def show_img(self, *_):
            pass

    def push(stack, row):
        return np.append(np.delete(stack, 0, 0), [row], 0)

    with rh.patched_choice(rec):
        env = rh.make_env(variant)
        et = env_buffer.Env_transact()
        et.reset(env, Win())
        sm9, st9 = et.predata_reset(env)
        a5, r5, d5, l5, f5 = np.zeros((5, 1)), np.zeros((5, 1)), np.zeros((5, 1)), np.zeros((5, 1)), np.zeros((5, 2))
        buf.epo_counter_start = buf.pre_counter

        def store():
            rec_ = (np.array(sm9, dtype='float32'), np.array(a5, dtype='int64'), np.array(st9, dtype='float32'),
                    np.array(goal, dtype='float32'), np.array(r5, dtype='float32'), np.array(d5, dtype='int64'),
                    np.array(l5, dtype='float32'))
            buf.pre_store(rec_ + ((np.array(f5, dtype='float32'),) if with_f else ()))
        for t in range(10000):
            a_idx = int(rs.choice(5, p=(0.1, 0.3, 0.35, 0.1, 0.15)))
            logp = float(np.log(0.2) - 0.01 * rs.rand())
            fut = [float(rs.randint(-3, 4)), float(rs.randint(-3, 4))]
            slots.begin_step(t)
            _, r, term, trunc, done = et.step(env, None, et.env_action(env, a_idx), Args)
            state, goal = et.data_env(env)
            sm = et.matrix_env(env)
            for k in range(5 if (term or trunc) else 1):           # the step itself + four terminal repeats
                st9, sm9 = push(st9, state), push(sm9, sm)
                a5, r5, d5, l5, f5 = push(a5, [a_idx]), push(r5, [r]), push(d5, [done]), push(l5, [logp]), push(f5, fut)
                if t > 3 or k > 0:
                    store()
            if term or trunc:
                return t + 1


def gen_window_her():
    """Buffer_gridworld.pre_her_func / pre_f_her_func (soa/env_buffer.py:145-280) on real episodes stored as 9-frame
    window records, incl. ring wrap; picks from the seeded global numpy stream."""
    env_buffer, _ = rh.soa_modules()
    out = {}
    cases = [dict(cap=320, seed=0, pre=0, variant="v6", f=False), dict(cap=320, seed=1, pre=23, variant="v4", f=False),
             dict(cap=150, seed=2, pre=60, variant="v6", f=False), dict(cap=320, seed=3, pre=5, variant="v4", f=True),
             dict(cap=140, seed=4, pre=70, variant="v6", f=True), dict(cap=320, seed=5, pre=0, variant="v6", f=True)]
    for ci, c in enumerate(cases):
        fields = [('s', np.float64, (9, 289)), ('a', np.int64, (5, 1)), ('p', np.float64, (9, 2)), ('g', np.float64, (2,)),
                  ('r', np.float64, (5, 1)), ('d', np.int64, (5, 1)), ('a_logp', np.float64, (5, 1))]
        if c["f"]:
            fields.append(('f', np.float64, (5, 2)))
        buf = env_buffer.Buffer_gridworld()
        buf.grid_size = 17
        buf.buffer_pre_capacity = c["cap"]
        buf.pre_transition = np.dtype(fields)
        buf.pre_buffer = np.zeros(c["cap"], dtype=buf.pre_transition)
        buf.pre_counter = c["pre"]
        L = collect_window_episode(env_buffer, c["variant"], buf, 40 + ci, c["f"])
        before = buf.pre_buffer.copy()
        cnt_before, full_before = buf.pre_counter, buf.pre_full
        np.random.seed(c["seed"])
        (buf.pre_f_her_func if c["f"] else buf.pre_her_func)(max_steps=50, newgoal_size_in=4)
        for k in buf.pre_transition.names:
            out["c%d_before_%s" % (ci, k)] = before[k].astype(np.float32) if k == "s" else before[k]
            out["c%d_after_%s" % (ci, k)] = buf.pre_buffer[k].astype(np.float32) if k == "s" else buf.pre_buffer[k]
            if k == "s":                            # frames were stored through float32 (pre_store call sites): lossless
                assert np.array_equal(before[k], before[k].astype(np.float32).astype(np.float64))
        out["c%d_meta" % ci] = np.array([c["cap"], c["seed"], c["pre"], L, cnt_before, int(full_before), buf.pre_counter,
                                         int(buf.pre_full), buf.epo_counter_end, int(c["f"])])
        print("  case %d: episode of %d steps, %d records before, counter %d -> %d%s" %
              (ci, L, (cnt_before - c["pre"]) % c["cap"], cnt_before, buf.pre_counter, " (wrapped)" if buf.pre_full else ""))
    out["n_cases"] = np.int32(len(cases))
    path_out = os.path.join(GOLD, "window_her.npz")
    np.savez_compressed(path_out, **out)
    print("window_her: %d cases -> %s (%.1f KB)" % (len(cases), path_out, os.path.getsize(path_out) / 1024))



def det_weights_v2(module, seed):
    """det_weights that also gives BatchNorm sane running statistics (var > 0) and keeps integer buffers."""
    import torch
    sd = det_weights(module, seed)
    for k, (name, prm) in enumerate(module.state_dict().items()):
        n = prm.numel()
        if name.endswith("num_batches_tracked"):
            sd[name] = prm.clone()
        elif name.endswith("running_var"):
            sd[name] = torch.tensor(1.0 + 0.3 * np.sin(0.37 * np.arange(n) + k + seed), dtype=prm.dtype)
        elif name.endswith("running_mean"):
            sd[name] = torch.tensor(0.1 * np.sin(0.53 * np.arange(n) + k + seed), dtype=prm.dtype)
        elif ".1.weight" in name or ".4.weight" in name or ".7.weight" in name:
            if prm.dim() == 1:                      # BatchNorm gamma
                sd[name] = torch.tensor(1.0 + 0.2 * np.sin(0.41 * np.arange(n) + k + seed), dtype=prm.dtype)
    return sd


def gen_predictor():
    """soa/agent/PPO_Predictor.py:72-83 pred_states + 8-frame actor/critic with seeded weights
    (the author's checkpoints are not in the repository)."""
    import torch
    env_buffer, ppo_mod = rh.soa_modules()
    from agent import PPO_Predictor as pp_mod
    pp_mod.heatmap = lambda *a, **k: None
    out = {}
    torch.manual_seed(SEED)
    agent = pp_mod.ppo_predictor()
    for tag, net in (("actor", agent.actor), ("critic", agent.critic), ("encoder", agent.encoder),
                     ("decoder", agent.decoder)):
        st = param_stats(net)
        out["init_%s_names" % tag] = np.array([x[0] for x in st])
        out["init_%s_sum" % tag] = np.array([x[2] for x in st])
        out["init_%s_abs" % tag] = np.array([x[3] for x in st])
    out["init_predictor_names"] = np.array(list(agent.predictor.state_dict().keys()))
    out["init_predictor_sum"] = np.array([float(v.double().sum()) for v in agent.predictor.state_dict().values()])
    for i, net in enumerate((agent.actor, agent.critic, agent.encoder, agent.decoder)):
        net.load_state_dict(det_weights_v2(net, 11 + i))
    lstm_sd = {}
    for k, (name, prm) in enumerate(agent.predictor.state_dict().items()):
        n = prm.numel()
        lstm_sd[name] = torch.tensor((0.03 * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.3 * k)).reshape(tuple(prm.shape)),
                                     dtype=prm.dtype)
    agent.predictor.load_state_dict(lstm_sd)
    dev = torch.device("cpu")
    agent.encoder.device = agent.predictor.device = dev
    buf = collect_buffer(env_buffer, "v4", 64, seed=4)
    b = buf.buffer
    s = torch.tensor(b['s'][[5, 20, 41]][:, 1:5])
    pp = torch.tensor(b['p'][[5, 20, 41]][:, 1:5]); g = torch.tensor(b['g'][[5, 20, 41]])
    heads, up, full = agent.pred_states(s)
    cat = torch.cat([s, heads.detach()], 1)
    agent.actor.eval(); agent.critic.eval()
    with torch.no_grad():
        out["probs"] = agent.actor(cat, pp, g).numpy()
        out["value"] = agent.critic(cat, pp, g).numpy()
    out["in_s"] = s.numpy(); out["in_p"] = pp.numpy(); out["in_g"] = g.numpy()
    out["pred_frames"] = heads.numpy()
    out["pred_full_sum"] = full.double().sum(dim=(2, 3, 4)).numpy()
    path_out = os.path.join(GOLD, "predictor.npz")
    np.savez_compressed(path_out, **out)
    print("predictor: pred_frames range [%.4f, %.4f] -> %s (%.1f KB)" % (heads.min(), heads.max(), path_out,
                                                                         os.path.getsize(path_out) / 1024))



def gen_predictor_update():
    """soa/agent/PPO_Predictor.py:123-193 ppo_predictor.update on 9-frame window records (train_ppo_predictor.py:105-107)
    with seeded weights: the per-minibatch losses, and the minibatch order it drew."""
    import torch
    env_buffer, ppo_mod = rh.soa_modules()
    from agent import PPO_Predictor as pp_mod
    pp_mod.heatmap = lambda *a, **k: None
    out = {}
    torch.manual_seed(SEED)
    agent = pp_mod.ppo_predictor()
    for i, net in enumerate((agent.actor, agent.critic, agent.encoder, agent.decoder)):
        net.load_state_dict(det_weights_v2(net, 11 + i))
    lstm_sd = {}
    for k, (name, prm) in enumerate(agent.predictor.state_dict().items()):
        n = prm.numel()
        lstm_sd[name] = torch.tensor((0.03 * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.3 * k)).reshape(tuple(prm.shape)),
                                     dtype=prm.dtype)
    agent.predictor.load_state_dict(lstm_sd)
    dev = torch.device("cpu")
    agent.encoder.device = agent.predictor.device = dev
    # window records of a real episode, stored and relabelled by the reference's own buffer code
    buf = env_buffer.Buffer_gridworld()
    buf.grid_size = 17
    buf.buffer_pre_capacity = 200
    buf.pre_transition = np.dtype([('s', np.float64, (9, 289)), ('a', np.int64, (5, 1)), ('p', np.float64, (9, 2)),
                                   ('g', np.float64, (2,)), ('r', np.float64, (5, 1)), ('d', np.int64, (5, 1)),
                                   ('a_logp', np.float64, (5, 1))])
    buf.pre_buffer = np.zeros(200, dtype=buf.pre_transition)
    collect_window_episode(env_buffer, "v4", buf, 77, False)
    np.random.seed(3)
    buf.pre_her_func(max_steps=50, newgoal_size_in=4)
    n = buf.pre_counter
    pb = buf.pre_buffer[np.linspace(0, n - 1, 48).astype(int)].copy()        # 48 records: real and hindsight ones
    agent.batch_size, agent.K_epochs = 16, 2
    agent.heatmapfilename = "x"
    torch.manual_seed(135)
    st0 = torch.get_rng_state()
    agent.update(pb, dev, 0)
    out["action_loss"] = np.array([v for _, v in agent.writer.scalars["loss/action_loss_update"]])
    out["value_loss"] = np.array([v for _, v in agent.writer.scalars["loss/value_loss_update"]])
    torch.set_rng_state(st0)
    out["perms"] = np.stack([torch.randperm(48).numpy() for _ in range(2)])
    for tag, net in (("actor", agent.actor), ("critic", agent.critic)):
        out["upd_%s_sum" % tag] = np.array([x[2] for x in param_stats(net)])
    for k in buf.pre_transition.names:
        out["buf_" + k] = pb[k].astype(np.float32) if k == "s" else pb[k]
    path_out = os.path.join(GOLD, "predictor_update.npz")
    np.savez_compressed(path_out, **out)
    print("predictor_update: %d of %d records (%d relabelled done flags), losses" % (48, n, int(pb['d'][:, 0].sum())),
          out["action_loss"][:2], out["value_loss"][:2], "-> %s (%.1f KB)" % (path_out, os.path.getsize(path_out) / 1024))


# ----------------------------------------------------------------------------- self-orientation agent
def gen_soa():
    """soa/agent/Self_orientation_agent.py: init statistics of the three trainable nets, forward of the orientation
    head / actor / critic on 8-frame inputs, and the per-minibatch losses of update_policy (:166-239) and
    update_orientation (:242-294) on 9-frame window records (train_SoA.py:113-117) with seeded weights."""
    import torch
    env_buffer, _ = rh.soa_modules()
    from agent import Self_orientation_agent as so_mod
    so_mod.heatmap = lambda *a, **k: None
    so_mod.savetxt = lambda *a, **k: None                 # update_orientation dumps a csv next to its logs
    out = {}
    torch.manual_seed(SEED)
    agent = so_mod.self_orinetation_agent()
    nets = (("actor", agent.actor), ("critic", agent.critic), ("orient", agent.agent_position_preditor))
    for tag, net in nets:
        st = param_stats(net)
        out["init_%s_names" % tag] = np.array([x[0] for x in st])
        out["init_%s_sum" % tag] = np.array([x[2] for x in st])
        out["init_%s_abs" % tag] = np.array([x[3] for x in st])
    for i, (tag, net) in enumerate(nets):
        net.load_state_dict(det_weights(net, 21 + i))
    for i, net in enumerate((agent.encoder, agent.decoder)):
        net.load_state_dict(det_weights_v2(net, 13 + i))
    lstm_sd = {}
    for k, (name, prm) in enumerate(agent.predictor.state_dict().items()):
        n = prm.numel()
        lstm_sd[name] = torch.tensor((0.03 * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.3 * k)).reshape(tuple(prm.shape)),
                                     dtype=prm.dtype)
    agent.predictor.load_state_dict(lstm_sd)
    dev = torch.device("cpu")
    agent.encoder.device = agent.predictor.device = dev
    # a real trajectory: per-step frames / positions of the first long episode of a seeded random policy
    buf = collect_buffer(env_buffer, "v4", 128, seed=6)
    b = buf.buffer
    ends = buf.episode_ends
    e0 = next(i for i, e in enumerate(ends) if (e - (ends[i - 1] + 1 if i else 0)) >= 44)
    first = ends[e0 - 1] + 1 if e0 else 0
    L = 44
    frames = np.concatenate([b['s'][first][:4], b['s'][first:first + L, 4]])       # 4 reset frames + one frame per step
    poss = np.concatenate([b['p'][first][:4], b['p'][first:first + L, 4]])
    B = 32
    rs = np.random.RandomState(11)
    pre_t = np.dtype([('s', np.float64, (9, 289)), ('a', np.int64, (5, 1)), ('p', np.float64, (9, 2)), ('g', np.float64, (2,)),
                      ('r', np.float64, (5, 1)), ('d', np.int64, (5, 1)), ('a_logp', np.float64, (5, 1)), ('f', np.float64, (5, 2))])
    pb = np.zeros(B, dtype=pre_t)
    for i in range(B):
        pb['s'][i] = frames[i:i + 9]
        pb['p'][i] = poss[i:i + 9]
        pb['g'][i] = b['g'][first]
        pb['a'][i] = rs.randint(0, 5, (5, 1))
        pb['r'][i] = rs.choice([-0.01, -0.1, 0.2, 0.9], (5, 1))
        pb['a_logp'][i] = np.log(0.2) - 0.05 * rs.rand(5, 1)
        pb['f'][i] = rs.randint(-3, 4, (5, 2))
    # (1) forward
    sel = [0, 7, 19]
    s4 = torch.tensor(pb['s'][sel][:, :4], dtype=torch.float32)
    p4 = torch.tensor(pb['p'][sel][:, :4], dtype=torch.float32)
    g = torch.tensor(pb['g'][sel], dtype=torch.float32)
    f0 = torch.tensor(pb['f'][sel][:, 0], dtype=torch.float32)
    for m in (agent.actor, agent.critic, agent.agent_position_preditor):
        m.eval()
    with torch.no_grad():
        heads, _, _ = agent.pred_states(s4)
        x8 = torch.cat([s4, heads], 1)
        px, py = agent.agent_position_preditor(x8, p4, g)
        cg = torch.cat([g, f0], 1)
        out["fwd_px"], out["fwd_py"] = px.numpy(), py.numpy()
        out["fwd_probs"] = agent.actor(x8, p4, cg).numpy()
        out["fwd_value"] = agent.critic(x8, p4, cg).numpy()
    out["fwd_sel"] = np.array(sel)
    # (2) update_policy: 32 records, minibatch 16, 2 epochs
    agent.batch_size, agent.K_epochs = 16, 2
    agent.heatmapfilename = "x"
    torch.manual_seed(321)
    st0 = torch.get_rng_state()
    agent.update_policy(pb, dev, 0)
    out["pol_action_loss"] = np.array([v for _, v in agent.writer.scalars["loss/action_loss_update"]])
    out["pol_value_loss"] = np.array([v for _, v in agent.writer.scalars["loss/value_loss_update"]])
    torch.set_rng_state(st0)
    out["pol_perms"] = np.stack([torch.randperm(B).numpy() for _ in range(2)])
    # (3) update_orientation: same records, minibatch 16, 2 epochs
    agent.batch_size_pre_agent, agent.K_epochs_pre_agent_position = 16, 2
    agent.future3positionfilename = "/tmp/soa_golden_"
    torch.manual_seed(654)
    st1 = torch.get_rng_state()
    agent.update_orientation(pb, dev, 0)
    out["ori_loss"] = np.array([v for _, v in agent.writer.scalars["loss/future_3steps_loss_update"]])
    torch.set_rng_state(st1)
    out["ori_perms"] = np.stack([torch.randperm(B).numpy() for _ in range(2)])
    for tag, net in nets:
        st = param_stats(net)
        out["upd_%s_sum" % tag] = np.array([x[2] for x in st])
    for k in pre_t.names:
        out["buf_" + k] = pb[k]
    path_out = os.path.join(GOLD, "soa.npz")
    np.savez_compressed(path_out, **out)
    print("soa: policy losses", out["pol_action_loss"][:2], out["pol_value_loss"][:2], "orientation", out["ori_loss"][:2],
          "-> %s (%.1f KB)" % (path_out, os.path.getsize(path_out) / 1024))


# ----------------------------------------------------------------------------- offline predictor pipeline
def gen_pretrain():
    """soa/agent/encoder_LSTM_decoder.py: update_encoder_decoder (:95-180) and update_predictor (:182-295) on a small
    buffer of 9-frame window records -- per-update train / validation losses with lr 5e-4 as train_encoder_decoder.py
    :113-114 sets it, 2 epochs, batch 8, in-process DataLoader."""
    import torch
    env_buffer, _ = rh.soa_modules()
    from agent import encoder_LSTM_decoder as eld
    eld.savetxt = lambda *a, **k: None                     # the loops dump csv files to an absolute home path
    eld.tqdm = lambda x, *a, **k: x
    out = {}
    buf = collect_buffer(env_buffer, "v4", 160, seed=8)
    b = buf.buffer
    frames = np.concatenate([b['s'][0][:4], b['s'][:, 4]]).astype(np.float64)      # consecutive frames of the random policy
    B = 48
    pre_t = np.dtype([('s', np.float64, (9, 289)), ('a', np.int64, (5, 1)), ('p', np.float64, (9, 2)), ('g', np.float64, (2,)),
                      ('r', np.float64, (5, 1)), ('d', np.int64, (5, 1)), ('a_logp', np.float64, (5, 1))])
    pb = np.zeros(B, dtype=pre_t)
    for i in range(B):
        pb['s'][i] = frames[3 * i:3 * i + 9]
    out["buf_s"] = pb['s'].astype(np.float32)
    torch.manual_seed(SEED)
    m = eld.encoder_lstm_decoder()
    for tag, net in (("encoder", m.encoder), ("decoder", m.decoder)):
        out["init_%s_sum" % tag] = np.array([x[2] for x in param_stats(net)])
    for i, net in enumerate((m.encoder, m.decoder)):
        net.load_state_dict(det_weights_v2(net, 31 + i))
    lstm_sd = {}
    for k, (name, prm) in enumerate(m.predictor.state_dict().items()):
        n = prm.numel()
        lstm_sd[name] = torch.tensor((0.03 * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.3 * k)).reshape(tuple(prm.shape)),
                                     dtype=prm.dtype)
    m.predictor.load_state_dict(lstm_sd)
    dev = torch.device("cpu")
    m.encoder.device = m.predictor.device = dev
    m.batch_size, m.num_workers, m.num_episodes_en_de, m.num_episodes_pre = 8, 0, 2, 2
    m.name = "golden"
    m.save_param = lambda *a, **k: None
    m.save_param_encoder_decoder = lambda *a, **k: None
    adam = lambda net: torch.optim.Adam(net.parameters(), lr=5e-04, betas=(0.9, 0.98), eps=1e-09)   # noqa: E731
    m.optimizer_encoder, m.optimizer_decoder, m.optimizer_predictor = adam(m.encoder), adam(m.decoder), adam(m.predictor)
    step = lambda o: torch.optim.lr_scheduler.StepLR(o, step_size=1, gamma=0.9)                      # noqa: E731
    m.scheduler_encoder, m.scheduler_decoder, m.scheduler_predictor = (step(m.optimizer_encoder), step(m.optimizer_decoder),
                                                                       step(m.optimizer_predictor))
    pb32 = np.zeros(B, dtype=np.dtype([('s', np.float32, (9, 289))]))
    pb32['s'] = pb['s']
    torch.manual_seed(111)
    m.update_encoder_decoder(pb32, dev)
    out["ed_train"] = np.array([v for _, v in m.en_de_writer.scalars["loss/en_de_train_loss_update"]])
    out["ed_val"] = np.array([v for _, v in m.en_de_writer.scalars["loss/en_de_value_loss_update"]])
    torch.manual_seed(222)
    m.update_predictor(pb32, dev)
    out["pre_train"] = np.array([v for _, v in m.writer.scalars["loss/pre_train_loss_update"]])
    out["pre_val"] = np.array([v for _, v in m.writer.scalars["loss/pre_value_loss_update"]])
    out["final_encoder_sum"] = np.array([x[2] for x in param_stats(m.encoder)])
    out["final_predictor_sum"] = np.array([float(v.double().sum()) for v in m.predictor.state_dict().values()])
    path_out = os.path.join(GOLD, "pretrain.npz")
    np.savez_compressed(path_out, **out)
    print("pretrain: en/de", out["ed_train"][:3], out["ed_val"][:2], "predictor", out["pre_train"][:3], out["pre_val"][:2],
          "-> %s (%.1f KB)" % (path_out, os.path.getsize(path_out) / 1024))


# ----------------------------------------------------------------------------- general MiniGrid views
def gen_occlusion():
    """MiniGridEnv.gen_obs / gen_obs_grid with see_through_walls False and True on random W x H grids holding
    every object class of the reference (walls, doors in all three states, keys, balls, boxes, goal, lava,
    floor), all four agent directions, odd view sizes 3..11, with and without a carried object
    (minigrid.py:1443-1496, process_vis :795-832, slice/rotate_left :627-660, encode :749-772)."""
    rh.setup()
    import gym_minigrid.minigrid as mg
    rs = np.random.RandomState(4242)
    env = rh.make_env("v6").unwrapped
    colors = list(mg.COLOR_TO_IDX.keys())

    def rand_obj():
        k = rs.randint(0, 100)
        c = colors[rs.randint(len(colors))]
        if k < 40:
            return None
        if k < 58:
            return mg.Wall()
        if k < 70:
            st = rs.randint(3)
            return mg.Door(c, is_open=(st == 0), is_locked=(st == 2))
        if k < 76:
            return mg.Key(c)
        if k < 82:
            return mg.Ball(c)
        if k < 88:
            return mg.Box(c)
        if k < 92:
            return mg.Goal()
        if k < 96:
            return mg.Lava()
        return mg.Floor(c)

    out = {}
    ncase = 0
    for (W, H) in [(17, 17), (9, 13), (25, 6), (5, 5), (12, 20)]:
        for rep in range(6):
            grid = mg.Grid(W, H)
            for j in range(H):
                for i in range(W):
                    grid.set(i, j, rand_obj())
            env.grid = grid
            env.width, env.height = W, H
            ax, ay = int(rs.randint(W)), int(rs.randint(H))
            env.agent_pos = (ax, ay)
            env.agent_dir = int(rs.randint(4))
            env.carrying = [None, mg.Key("blue"), mg.Ball("red"), mg.Box("yellow")][rs.randint(4)] if rep % 2 else None
            V = int([3, 5, 7, 9, 11, 7][rep])
            env.agent_view_size = V
            enc = grid.encode().astype(np.uint8)                      # [W][H][3] world planes
            for st in (0, 1):
                env.see_through_walls = bool(st)
                obs = env.gen_obs()
                _, vis = env.gen_obs_grid()
                out["c%03d_img%d" % (ncase, st)] = obs["image"].astype(np.uint8)
                out["c%03d_vis%d" % (ncase, st)] = vis.astype(np.uint8)
            out["c%03d_grid" % ncase] = enc
            carry = env.carrying.encode() if env.carrying is not None else (0, 0, 0)
            out["c%03d_meta" % ncase] = np.array([W, H, ax, ay, env.agent_dir, V, int(env.carrying is not None)]
                                                 + list(carry), np.int32)
            ncase += 1
    out["n_cases"] = np.int32(ncase)
    path_out = os.path.join(GOLD, "occlusion.npz")
    np.savez_compressed(path_out, **out)
    print("occlusion: %d cases -> %s (%.1f KB)" % (ncase, path_out, os.path.getsize(path_out) / 1024))


def gen_mgstep():
    """MiniGridEnv.step (the base-class transition, minigrid.py:1333-1441) on random worlds: absolute moves with the
    can_overlap rules of every object class, goal termination with _reward() (:1061), truncation, and the
    exceptions of out-of-range cells / actions outside {left,right,up,down,done}."""
    rh.setup()
    import gym_minigrid.minigrid as mg
    rs = np.random.RandomState(777)
    env = rh.make_env("v6").unwrapped
    colors = list(mg.COLOR_TO_IDX.keys())
    makers = [lambda c: None, lambda c: None, lambda c: mg.Wall(), lambda c: mg.Door(c, is_open=True),
              lambda c: mg.Door(c), lambda c: mg.Door(c, is_locked=True), lambda c: mg.Key(c), lambda c: mg.Ball(c),
              lambda c: mg.Box(c), lambda c: mg.Goal(), lambda c: mg.Lava(), lambda c: mg.Floor(c), lambda c: mg.SubGoal()]
    out = {}
    ncase = 0
    for (W, H) in [(7, 7), (5, 9), (12, 4)]:
        for rep in range(4):
            grid = mg.Grid(W, H)
            for j in range(H):
                for i in range(W):
                    grid.set(i, j, makers[rs.randint(len(makers))](colors[rs.randint(len(colors))]))
            env.grid, env.width, env.height = grid, W, H
            env.max_steps = int([50, 20, 9, 33][rep])
            env.see_through_walls = True
            env.agent_view_size = 3
            env.carrying = None
            rows = []
            env.agent_pos = (int(rs.randint(W)), int(rs.randint(H)))
            env.agent_dir = int(rs.randint(4))
            env.step_count = 0
            for t in range(60):
                if rs.rand() < 0.15:                       # teleport: visit borders and every object class
                    env.agent_pos = (int(rs.randint(W)), int(rs.randint(H)))
                    env.agent_dir = int(rs.randint(4))
                a = int(rs.choice([0, 1, 2, 3, 6, 0, 1, 2, 3, 6, 4, 5, 7, -1]))
                before = (int(env.agent_pos[0]), int(env.agent_pos[1]), int(env.agent_dir), int(env.step_count))
                err, r, te, tr = 0, 0.0, False, False
                try:
                    _, r, te, tr, _ = mg.MiniGridEnv.step(env, a)
                except AttributeError:
                    err = 1
                except AssertionError:
                    err = 2
                except ValueError:
                    err = 4
                rows.append(before + (a, int(env.agent_pos[0]), int(env.agent_pos[1]), int(env.step_count), err,
                                      int(te), int(tr)) + (float(r),))
            out["c%02d_grid" % ncase] = grid.encode().astype(np.uint8)
            out["c%02d_rows" % ncase] = np.array(rows, np.float64)
            out["c%02d_meta" % ncase] = np.array([W, H, env.max_steps], np.int32)
            ncase += 1
    out["n_cases"] = np.int32(ncase)
    path_out = os.path.join(GOLD, "mgstep.npz")
    np.savez_compressed(path_out, **out)
    print("mgstep: %d cases -> %s (%.1f KB)" % (ncase, path_out, os.path.getsize(path_out) / 1024))


STAGES = {"traces": gen_traces, "views": gen_views, "ppo": gen_ppo, "her": gen_her, "window_her": gen_window_her, "predictor": gen_predictor, "predictor_update": gen_predictor_update, "occlusion": gen_occlusion, "mgstep": gen_mgstep, "soa": gen_soa, "pretrain": gen_pretrain}


def main(argv):
    os.makedirs(GOLD, exist_ok=True)
    want = argv or list(STAGES)
    for name in want:
        STAGES[name]()


if __name__ == "__main__":
    main(sys.argv[1:])
