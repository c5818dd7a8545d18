"""Offline world-model training vs the per-update losses recorded from the reference's own
encoder_lstm_decoder.update_encoder_decoder / update_predictor (tests/golden/pretrain.npz).  Pure torch: CPU."""
import numpy as np
import torch

from test_ppo_common import GOLDEN
from test_predictor_cpu import det_weights_v2


def check_offline_world_model_training(device, atol, atol_first=None, n_first=2, atol_pre=None):
    """update_encoder_decoder / update_predictor (encoder_LSTM_decoder.py:95-290) on `device`: every train and
    validation loss of both stages vs the losses the reference logged on the same 48 records."""
    from twoarmy_amd.soa.agent.encoder_LSTM_decoder import encoder_lstm_decoder
    g = dict(np.load(GOLDEN + "/pretrain.npz"))
    torch.manual_seed(9981)
    m = encoder_lstm_decoder()
    for tag, net in (("encoder", m.encoder), ("decoder", m.decoder)):
        assert np.array_equal(np.array([float(v.double().sum()) for v in net.state_dict().values()]),
                              g["init_%s_sum" % tag]), tag
    for i, net in enumerate((m.encoder, m.decoder)):
        net.load_state_dict(det_weights_v2(net, 31 + i))
    sd = {}
    for k, (name, prm) in enumerate(m.predictor.state_dict().items()):
        n = prm.numel()
        sd[name] = torch.tensor((0.03 * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.3 * k)).reshape(tuple(prm.shape)),
                                dtype=prm.dtype)
    m.predictor.load_state_dict(sd)
    m.batch_size, m.num_episodes_en_de, m.num_episodes_pre, m.save_every, m.name = 8, 2, 2, 0, "t"
    adam = lambda net: torch.optim.Adam(net.parameters(), lr=5e-04, betas=(0.9, 0.98), eps=1e-09)       # noqa: E731
    m.optimizer_encoder, m.optimizer_decoder, m.optimizer_predictor = adam(m.encoder), adam(m.decoder), adam(m.predictor)
    step = lambda o: torch.optim.lr_scheduler.StepLR(o, step_size=1, gamma=0.9)                          # noqa: E731
    m.scheduler_encoder, m.scheduler_decoder = step(m.optimizer_encoder), step(m.optimizer_decoder)
    m.scheduler_predictor = step(m.optimizer_predictor)
    buf = np.zeros(g["buf_s"].shape[0], dtype=np.dtype([("s", np.float32, (9, 289))]))
    buf["s"] = g["buf_s"]
    torch.manual_seed(111)
    m.update_encoder_decoder(buf, device)
    sc = m.en_de_writer.scalars
    if atol_first is not None:          # before optimiser-step differences have compounded: the tight tolerance
        np.testing.assert_allclose([v for _, v in sc["loss/en_de_train_loss_update"]][:n_first], g["ed_train"][:n_first],
                                   rtol=0, atol=atol_first)
    np.testing.assert_allclose([v for _, v in sc["loss/en_de_train_loss_update"]], g["ed_train"], rtol=0, atol=atol)
    np.testing.assert_allclose([v for _, v in sc["loss/en_de_value_loss_update"]], g["ed_val"], rtol=0, atol=atol)
    torch.manual_seed(222)
    m.update_predictor(buf, device)
    sc = m.writer.scalars
    if atol_first is not None:
        np.testing.assert_allclose([v for _, v in sc["loss/pre_train_loss_update"]][:n_first], g["pre_train"][:n_first],
                                   rtol=0, atol=atol_first * 10)     # the stage starts from stage 1's (drifted) encoder
    np.testing.assert_allclose([v for _, v in sc["loss/pre_train_loss_update"]], g["pre_train"], rtol=0, atol=atol_pre or atol)
    np.testing.assert_allclose([v for _, v in sc["loss/pre_value_loss_update"]], g["pre_val"], rtol=0, atol=atol_pre or atol)
    if atol <= 1e-5:        # parameter checksums only for the bit-reproducible (CPU) run: Adam(eps=1e-9) steps every
        #                     near-zero-gradient element by +-lr, so sums over 4 M elements are chaotic across hardware
        np.testing.assert_allclose([float(v.double().sum()) for v in m.predictor.state_dict().values()],
                                   g["final_predictor_sum"], rtol=1e-5, atol=1e-4)


def test_offline_world_model_training_matches_reference():
    check_offline_world_model_training("cpu", 1e-5)


def test_window_records_equal_literal_window_shifts():
    """datacol_predictor.windows_from_rollout (index arithmetic) vs the reference's literal 9-deep np.delete /
    np.append stacks with the four terminal shifts (datacol_predictor.py:118-163), on an oracle rollout."""
    import twoarmy_oracle as orc
    from twoarmy_amd.soa.datacol_predictor import windows_from_rollout
    N, T = 24, 110
    ref = orc.rollout(4, N, T, 9981, view=3, want_obs=False)
    frames, pos = ref["matrix"], ref["pos"]
    term, trunc = ref["terminated"] != 0, ref["truncated"] != 0
    init_f = np.asarray(orc.OracleEnv(4).matrix(), np.float32)
    init_p = np.array([15.0, 3.0], np.float32)
    a = (np.arange(T * N, dtype=np.int32).reshape(T, N) * 7) % 5
    w = windows_from_rollout(torch.tensor(frames), torch.tensor(pos), torch.tensor(a), torch.tensor(ref["reward"]),
                             torch.tensor(ref["terminated"]), torch.tensor(ref["truncated"]), torch.tensor(init_f))
    got = {(int(t), int(n)): i for i, (t, n) in enumerate(zip(w["t"], w["n"]))}
    S, P, A = w["s"].numpy(), w["p"].numpy(), w["a"].numpy()
    checked = 0
    for n in range(N):
        s9 = np.tile(init_f, (9, 1)); p9 = np.tile(init_p, (9, 1)); a5 = np.zeros(5, np.int64); steps = []
        for t in range(T):
            s9 = np.append(np.delete(s9, 0, 0), [frames[t, n]], 0); p9 = np.append(np.delete(p9, 0, 0), [pos[t, n]], 0)
            a5 = np.append(np.delete(a5, 0), a[t, n]); steps.append(t)
            recs = []
            if len(steps) > 4:
                recs.append((steps[-5], s9.copy(), p9.copy(), a5.copy()))
            if term[t, n] or trunc[t, n]:
                for sh in range(4):
                    s9 = np.append(np.delete(s9, 0, 0), [frames[t, n]], 0); p9 = np.append(np.delete(p9, 0, 0), [pos[t, n]], 0)
                    a5 = np.append(np.delete(a5, 0), a[t, n])
                    if len(steps) >= 4 - sh:
                        recs.append((steps[-4 + sh], s9.copy(), p9.copy(), a5.copy()))
                s9 = np.tile(init_f, (9, 1)); p9 = np.tile(init_p, (9, 1)); a5 = np.zeros(5, np.int64); steps = []
            for (tau, rs, rp, ra) in recs:
                i = got[(tau, n)]
                assert np.array_equal(S[i], rs) and np.array_equal(P[i], rp) and np.array_equal(A[i], ra), (tau, n)
                checked += 1
    assert checked == len(got) and checked > 2000
