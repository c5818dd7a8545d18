"""Entry point of the auto-encoder stage of the offline world-model training (reference
soa/train_encoder_decoder.py:53-123): loads a window-record buffer written by datacol_predictor and runs
encoder_lstm_decoder.update_encoder_decoder with Adam(5e-4, betas (0.9, 0.98), eps 1e-9) and StepLR(1, 0.9).

  python -m twoarmy_amd.soa.train_encoder_decoder --buffer_file out/predictor_....npy --num_episodes 10
"""
import argparse
import os
import random

import numpy as np
import torch
from torch.optim.lr_scheduler import StepLR

from .agent.encoder_LSTM_decoder import encoder_lstm_decoder


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--seed", type=int, default=3344)
    p.add_argument("--batch_size", type=int, default=2)
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--lr", type=float, default=5e-04)
    p.add_argument("--num_episodes", type=int, default=10000)
    p.add_argument("--buffer_file", required=True)
    p.add_argument("--log_dir", default=None)
    p.add_argument("--cuda", default="cuda:0")
    return p


def seed_everything(seed):
    random.seed(seed); np.random.seed(seed); os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None):
    args = build_parser().parse_args(argv)
    device = torch.device(args.cuda if torch.cuda.is_available() else "cpu")
    seed_everything(args.seed)
    buffer = np.load(args.buffer_file)                  # plain structured array: no pickle involved
    m = encoder_lstm_decoder(log_root=args.log_dir)
    m.seed, m.num_episodes_en_de, m.batch_size, m.num_workers = args.seed, args.num_episodes, args.batch_size, args.num_workers
    m.name = "MiniGrid-twoarmy-17x17_encoder_decoder_"
    m.encoder.to(device); m.decoder.to(device)
    adam = lambda net: torch.optim.Adam(net.parameters(), lr=args.lr, betas=(0.9, 0.98), eps=1e-09)     # noqa: E731
    m.optimizer_encoder, m.optimizer_decoder = adam(m.encoder), adam(m.decoder)
    m.scheduler_encoder, m.scheduler_decoder = StepLR(m.optimizer_encoder, 1, 0.9), StepLR(m.optimizer_decoder, 1, 0.9)
    tl, vl = m.update_encoder_decoder(buffer, device)
    print("update over: train %.6f val %.6f" % (tl, vl))
    return m


if __name__ == "__main__":
    main()
