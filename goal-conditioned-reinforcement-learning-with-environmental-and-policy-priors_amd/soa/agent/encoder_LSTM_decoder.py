"""Offline training of the world model (reference soa/agent/encoder_LSTM_decoder.py:23-295): the per-frame
encoder / decoder pair is trained as an auto-encoder on single frames, then the LSTM is trained to roll the latents of
4 frames 4 steps ahead with encoder and decoder frozen.  Same attribute / method names, scalar tags and checkpoint
keys as the reference; the data are 9-frame window records (`datacol_predictor.collect_windows`).

Everything here is PyTorch-ROCm (convs, LSTM GEMMs on MFMA); the split and shuffling use the same library calls as
the reference (sklearn train_test_split(random_state=1), torch DataLoader(shuffle=True)), so a seeded run visits the
same minibatches."""
import os
from datetime import datetime
from itertools import chain
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

from .net.all_net import LSTM, Net_Decoder, Net_Encoder
from .PPO import _make_writer


class encoder_lstm_decoder:
    def __init__(self, log_root=None):
        self.traindate = datetime.now().strftime("%Y_%m_%d_%H_%M_%S")
        root = Path(log_root or os.environ.get("TWOARMY_LOG_ROOT", "./runs"))
        self.filepath = root / "param" / "ppo_encoder_decoder_predictor" / ("gridim_pre" + self.traindate)
        self.writer_dir = str(root / "tensor" / "predictor" / ("logs_" + self.traindate))
        self.writer = _make_writer(self.writer_dir)
        self.en_de_filepath = root / "param" / "ppo_encoder_decoder" / ("gridim_encoder_decoder" + self.traindate)
        self.en_de_writer_dir = str(root / "tensor" / "encoder_decoder" / ("logs_" + self.traindate))
        self.en_de_writer = _make_writer(self.en_de_writer_dir)
        self.encoder = Net_Encoder()
        self.decoder = Net_Decoder()
        self.predictor = LSTM()
        self.loss_func = nn.MSELoss(reduction="none")
        self.all_params = chain(self.encoder.parameters(), self.decoder.parameters(), self.predictor.parameters())
        self.gamma, self.lr = 0.99, 0.0001
        self.encoder_lr = self.decoder_lr = self.predictor_lr = 1e-08
        self.weight_decay = 0.0001
        self.lr_step_size, self.lr_gamma = 1, 0.9
        self.batch_size = 128
        self.num_workers = 0                      # in-process loading: the data already sit in memory
        adam = lambda net, lr: torch.optim.Adam(net.parameters(), lr=lr, betas=(0.9, 0.98), eps=1e-09)      # noqa: E731
        self.optimizer_encoder = adam(self.encoder, self.encoder_lr)
        self.optimizer_decoder = adam(self.decoder, self.decoder_lr)
        self.optimizer_predictor = adam(self.predictor, self.predictor_lr)
        step = lambda o: torch.optim.lr_scheduler.StepLR(o, step_size=self.lr_step_size, gamma=self.lr_gamma)  # noqa: E731
        self.scheduler_encoder, self.scheduler_decoder = step(self.optimizer_encoder), step(self.optimizer_decoder)
        self.scheduler_predictor = step(self.optimizer_predictor)
        self.num_episodes_en_de = 5
        self.num_episodes_pre = 5
        self.train_update_number_en_de = self.val_number_en_de = 0
        self.train_update_number_pre = self.val_number_pre = 0
        self.seed = 0
        self.en_de_average_score = self.pre_average_score = 0
        self.save_every = 2
        self.name = None

    # ------------------------------------------------------------------ checkpoints (reference :73-92)
    def _save(self, path_dir, state, fname):
        os.makedirs(path_dir, exist_ok=True)
        path = os.path.join(str(path_dir), fname + datetime.now().strftime("%Y_%m_%d_%H_%M_%S") + ".pkl")
        torch.save(state, path)
        return path

    def save_param(self, i_ep):
        state = {"model_encoder": self.encoder.state_dict(), "model_decoder": self.decoder.state_dict(),
                 "model_predictor": self.predictor.state_dict(), "optimizer_encoder": self.encoder.state_dict(),
                 "optimizer_decoder": self.decoder.state_dict(), "optimizer_predictor": self.predictor.state_dict(),
                 "epoch": i_ep}                    # the reference stores the model dicts under optimizer_* as well (:74)
        return self._save(self.filepath, state, "%s_pre_net_%saverage_score_%depoch_%sseed_"
                          % (self.name, self.pre_average_score, i_ep, self.seed))

    def save_param_encoder_decoder(self, i_ep):
        state = {"model_encoder": self.encoder.state_dict(), "model_decoder": self.decoder.state_dict(),
                 "optimizer_encoder": self.optimizer_encoder.state_dict(),
                 "optimizer_decoder": self.optimizer_decoder.state_dict(), "epoch": i_ep}
        return self._save(self.en_de_filepath, state, "%s_net_%saverage_score_%depoch_%sseed_"
                          % (self.name, self.en_de_average_score, i_ep, self.seed))

    # ------------------------------------------------------------------ data
    def _loaders(self, array):
        from sklearn.model_selection import train_test_split
        s_train, s_val = train_test_split(array, test_size=0.1, random_state=1)
        mk = lambda a: DataLoader(TensorDataset(torch.tensor(a)), batch_size=self.batch_size, shuffle=True,   # noqa: E731
                                  num_workers=self.num_workers)
        val_loader = mk(s_val)                     # same creation order as the reference (:105-112)
        train_loader = mk(s_train)
        return train_loader, val_loader

    # ------------------------------------------------------------------ auto-encoder (reference :95-180)
    def _recon_loss(self, states):
        z, z_upsample = self.encoder(states)
        _, full = self.decoder(z)
        return self.loss_func(z_upsample, full).mean((2, 3, 4)).mean()

    def update_encoder_decoder(self, buffer, device):
        """buffer['s'][:, 4] (the newest-but-4 frame of every 9-frame record) -> single-frame reconstruction."""
        device = torch.device(device)
        train_loader, val_loader = self._loaders(np.asarray(buffer["s"][:, 4]).reshape(-1, 1, 289))
        for p in chain(self.encoder.parameters(), self.decoder.parameters()):
            p.requires_grad = True
        self.encoder.to(device); self.decoder.to(device)
        train_loss = val_loss = average_loss = 0.0
        self.encoder.train(); self.decoder.train()     # (the reference never switches back to train() after the first
        for i_ep in range(self.num_episodes_en_de):    #  validation pass either: later epochs train in eval mode, :165)
            for i, (states,) in enumerate(train_loader):
                loss = self._recon_loss(states.to(device).float())
                self.optimizer_encoder.zero_grad(); self.optimizer_decoder.zero_grad()
                loss.backward()
                self.optimizer_encoder.step(); self.optimizer_decoder.step()
                train_loss = loss.detach()
                self.en_de_writer.add_scalar("loss/en_de_train_loss_update", train_loss, self.train_update_number_en_de)
                self.train_update_number_en_de += 1
                if i % 100 == 99:
                    average_loss = average_loss * 0.99 + float(train_loss) * 0.01
            self.encoder.eval(); self.decoder.eval()
            with torch.no_grad():
                for (states,) in val_loader:
                    val_loss = self._recon_loss(states.to(device).float())
                    self.en_de_writer.add_scalar("loss/en_de_value_loss_update", val_loss, self.val_number_en_de)
                    self.val_number_en_de += 1
            self.scheduler_encoder.step(); self.scheduler_decoder.step()
            if self.save_every and (i_ep + 1) % self.save_every == 0:
                self.en_de_average_score = average_loss
                self.save_param_encoder_decoder(i_ep)
        return float(train_loss), float(val_loss)

    # ------------------------------------------------------------------ latent predictor (reference :182-295)
    def _predictor_loss(self, states):
        z_c, z_up = self.encoder(states.reshape(-1, 1, 289))
        z_c = z_c.view(-1, 9, 64, 4, 4)
        z_pred, _ = self.predictor(z_c[:, :4].detach())
        _, full = self.decoder(z_pred[:, 3:7])
        return self.loss_func(z_up.reshape(-1, 9, 1, 68, 68)[:, 4:8], full).mean((2, 3, 4)).mean()

    def update_predictor(self, buffer, device):
        """buffer['s'] (9 frames per record): frames 0..3 in, frames 4..7 are the reconstruction targets."""
        device = torch.device(device)
        train_loader, val_loader = self._loaders(np.asarray(buffer["s"]))
        for p in chain(self.encoder.parameters(), self.decoder.parameters()):
            p.requires_grad = False                    # encoder and decoder stay fixed
        for m in (self.encoder, self.decoder, self.predictor):
            m.to(device)
        train_loss = val_loss = pre_average_loss = 0.0
        for i_ep in range(self.num_episodes_pre):
            pre_average_loss = 0.0
            self.predictor.train()
            for i, (states,) in enumerate(train_loader):
                loss = self._predictor_loss(states.to(device).float())
                self.optimizer_predictor.zero_grad()
                loss.backward()
                self.optimizer_predictor.step()
                train_loss = loss.detach()
                self.writer.add_scalar("loss/pre_train_loss_update", train_loss, self.train_update_number_pre)
                self.train_update_number_pre += 1
                if i % 100 == 99:
                    pre_average_loss = pre_average_loss * 0.99 + float(train_loss) * 0.01
            self.predictor.eval()
            with torch.no_grad():
                for (states,) in val_loader:
                    val_loss = self._predictor_loss(states.to(device).float())
                    self.writer.add_scalar("loss/pre_value_loss_update", val_loss, self.val_number_pre)
                    self.val_number_pre += 1
            if i_ep > 1:
                self.scheduler_predictor.step()
                if self.save_every and i_ep % self.save_every == 0:
                    self.pre_average_score = pre_average_loss
                    self.save_param(i_ep)
        return float(train_loss), float(val_loss)
