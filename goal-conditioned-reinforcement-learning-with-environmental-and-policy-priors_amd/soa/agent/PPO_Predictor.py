"""PPO whose actor / critic see the 4 real frames plus 4 frames predicted by a frozen encoder -> LSTM ->
decoder world model (reference soa/agent/PPO_Predictor.py:72-193).  Only actor and critic train; the
loss / sampling / advantage math is the same HIP path as PPO (ppo_ops)."""
import os
from datetime import datetime

import numpy as np
import torch

from .net.all_net import (LSTM, Net_Decoder, Net_Encoder, Net_PPO_Predictor_actor, Net_PPO_Predictor_critic)
from .PPO import PPO


class ppo_predictor(PPO):
    # the entry points of this agent store 9-frame windows from the fifth step on (train_ppo_predictor.py:134,
    # train_SoA.py:164) and relabel them with pre_her_func / pre_f_her_func (env_buffer.py:145-280): goal candidates are
    # the states after steps 4, 5, ... of an episode.  VecPPOTrainer.relabel passes this to ppo_her_relabel_window.
    her_window_delay = 4

    def __init__(self, log_root=None, use_tensorboard=False):
        super().__init__(log_root=log_root, use_tensorboard=use_tensorboard)
        # same construction order as the reference (PPO_Predictor.py:32-36)
        self.actor = Net_PPO_Predictor_actor()
        self.critic = Net_PPO_Predictor_critic()
        self.encoder = Net_Encoder()
        self.decoder = Net_Decoder()
        self.predictor = LSTM()
        self.pred_chunk = 4096                     # samples per pass of the frozen encoder -> LSTM -> decoder
        self.optimizer_actor = torch.optim.Adam(self.actor.parameters(), lr=self.lr, eps=1e-5)
        self.optimizer_critic = torch.optim.Adam(self.critic.parameters(), lr=self.lr, eps=1e-5)
        self.scheduler_actor = torch.optim.lr_scheduler.StepLR(self.optimizer_actor, self.lr_step_size, self.lr_gamma)
        self.scheduler_critic = torch.optim.lr_scheduler.StepLR(self.optimizer_critic, self.lr_step_size, self.lr_gamma)

    def to(self, device):
        for m in (self.actor, self.critic, self.encoder, self.decoder, self.predictor):
            m.to(device)
        return self

    @torch.no_grad()
    def pred_states(self, state_matrix):
        """(B,4,289) -> predicted next 4 frames (B,4,289) (+ upsampled inputs, full-res predictions)."""
        self.encoder.eval(); self.decoder.eval(); self.predictor.eval()
        B = state_matrix.shape[0]
        if B <= self.pred_chunk:
            z_c, z_up = self.encoder(state_matrix.reshape(-1, 1, 289))
            z_pred, _ = self.predictor(z_c.view(-1, 4, 64, 4, 4))
            frames, full = self.decoder(z_pred[:, 3:7])
            return frames, z_up, full
        # the frozen world model in slices: MIOpen's RNN rejects the 3 x 1024 LSTM at 32 768 sequences
        # (miopenStatusBadParm), and nothing here carries a gradient or a batch statistic
        parts = [self.pred_states(state_matrix[i:i + self.pred_chunk]) for i in range(0, B, self.pred_chunk)]
        return tuple(torch.cat([p[k] for p in parts]) for k in range(3))

    @torch.no_grad()
    def pred_frames(self, state_matrix):
        """The predicted next 4 frames only (what policy_input needs): the world model without materialising the
        upsampled inputs, in slices of pred_chunk samples."""
        self.encoder.eval(); self.decoder.eval(); self.predictor.eval()
        out = []
        for i in range(0, state_matrix.shape[0], self.pred_chunk):
            z_c, _ = self.encoder(state_matrix[i:i + self.pred_chunk].reshape(-1, 1, 289), need_upsampled=False)
            z_pred, _ = self.predictor(z_c.view(-1, 4, 64, 4, 4))
            out.append(self.decoder(z_pred[:, 3:7], need_full=False)[0])
        return out[0] if len(out) == 1 else torch.cat(out)

    def policy_input(self, frames4):
        """What actor / critic consume: the 4 real frames followed by the 4 predicted ones (8 channels)."""
        return torch.cat([frames4, self.pred_frames(frames4).detach()], dim=1)

    def _unpack(self, buffer, device):
        """9-frame window records (train_ppo_predictor.py:105-107): the update trains on transition 0 of a window --
        frames 0..4, slot 0 of a / r / a_logp (PPO_Predictor.py:126-155)."""
        f32 = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=device)
        a = torch.as_tensor(np.ascontiguousarray(buffer["a"][:, 0, 0]), dtype=torch.int64, device=device).to(torch.int32)
        return (f32(buffer["s"][:, :5]), f32(buffer["p"][:, :5]), a, f32(buffer["g"]), f32(buffer["r"][:, 0, 0]),
                f32(buffer["a_logp"][:, 0, 0]).view(-1, 1))

    def save_param(self, i_ep, running_score):
        """Checkpoint with the reference's keys (PPO_Predictor.py:113-120): actor / critic + the frozen world model.  The
        reference also stores optimiser states of the world model, which its update never steps; they are omitted."""
        state = {"model_actor": self.actor.state_dict(), "model_critic": self.critic.state_dict(),
                 "model_encoder": self.encoder.state_dict(), "model_decoder": self.decoder.state_dict(),
                 "model_predictor": self.predictor.state_dict(), "optimizer_actor": self.optimizer_actor.state_dict(),
                 "optimizer_critic": self.optimizer_critic.state_dict(), "epoch": i_ep}
        os.makedirs(self.filepath, exist_ok=True)
        path = os.path.join(str(self.filepath), "%s_net_%depoch_%srunning_score%s.pkl"
                            % (self.name, i_ep, running_score, datetime.now().strftime("%Y_%m_%d_%H_%M_%S")))
        torch.save(state, path)
        return path

    def load_world_model(self, checkpoint):
        """Checkpoint dict with the reference's keys 'model_encoder', 'model_decoder', 'model_predictor'
        (train_ppo_predictor.py:81-85)."""
        self.encoder.load_state_dict(checkpoint["model_encoder"])
        self.decoder.load_state_dict(checkpoint["model_decoder"])
        self.predictor.load_state_dict(checkpoint["model_predictor"])
