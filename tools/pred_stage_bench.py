#!/usr/bin/env python3
"""Where the frozen world model's time goes (PPO + predictor head, configs[4]): encoder / LSTM / decoder at one
rollout step's batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from twoarmy_amd.soa.agent.PPO_Predictor import ppo_predictor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
ag = ppo_predictor().to("cuda:0")
for m in (ag.encoder, ag.decoder, ag.predictor):
    m.eval()
s = torch.tensor([0.9, -0.9, -0.5, 0.3], device="cuda:0")[torch.randint(0, 4, (B, 4, 289), device="cuda:0")]


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


with torch.no_grad():
    t_enc, (z_c, z_up) = timed(lambda: ag.encoder(s.reshape(-1, 1, 289)))
    zc = z_c.view(-1, 4, 64, 4, 4)
    t_lstm, (z_pred, _) = timed(lambda: ag.predictor(zc))
    t_rnn, _ = timed(lambda: ag.predictor.recurrent_model(zc.reshape(B, 4, 1024)))
    t_dec, _ = timed(lambda: ag.decoder(z_pred[:, 3:7]))
    t_all, _ = timed(lambda: ag.pred_states(s))
    t_actor, _ = timed(lambda: ag.actor(ag.policy_input(s), torch.zeros(B, 4, 2, device="cuda:0"), torch.zeros(B, 2, device="cuda:0")))
print("B=%d  encoder %.2f ms  lstm(gemm path) %.2f ms  [miopen rnn, 4 known steps only: %.2f ms]  decoder %.2f ms  pred_states %.2f ms  policy_input+actor %.2f ms"
      % (B, t_enc, t_lstm, t_rnn, t_dec, t_all, t_actor))
