"""Entry point of the latent-predictor stage (reference soa/train_predictor.py:30-97): loads the window-record
buffer and the encoder / decoder checkpoint of train_encoder_decoder, freezes them and trains the LSTM with
Adam(1e-4, betas (0.9, 0.98), eps 1e-9) and StepLR(1, 0.9).  The checkpoint it writes holds the keys
'model_encoder' / 'model_decoder' / 'model_predictor' that train_ppo_predictor / train_SoA load (--predictor_file).

  python -m twoarmy_amd.soa.train_predictor --buffer_file out/predictor_....npy --net_file runs/.../..._net_....pkl
"""
import numpy as np
import torch
from torch.optim.lr_scheduler import StepLR

from .agent.encoder_LSTM_decoder import encoder_lstm_decoder
from .train_encoder_decoder import build_parser as _base_parser, seed_everything


def main(argv=None):
    p = _base_parser()
    p.set_defaults(lr=1e-04, seed=3344)
    p.add_argument("--net_file", default=None, help="encoder / decoder checkpoint of train_encoder_decoder")
    args = p.parse_args(argv)
    device = torch.device(args.cuda if torch.cuda.is_available() else "cpu")
    seed_everything(args.seed)
    buffer = np.load(args.buffer_file)
    m = encoder_lstm_decoder(log_root=args.log_dir)
    m.seed, m.num_episodes_pre, m.batch_size, m.num_workers = args.seed, args.num_episodes, args.batch_size, args.num_workers
    m.name = "MiniGrid-twoarmy-17x17_predictor_"
    if args.net_file:
        ck = torch.load(args.net_file, map_location="cpu", weights_only=True)
        m.encoder.load_state_dict(ck["model_encoder"]); m.decoder.load_state_dict(ck["model_decoder"])
    m.optimizer_predictor = torch.optim.Adam(m.predictor.parameters(), lr=args.lr, betas=(0.9, 0.98), eps=1e-09)
    m.scheduler_predictor = StepLR(m.optimizer_predictor, step_size=1, gamma=0.9)
    tl, vl = m.update_predictor(buffer, device)
    print("update over: train %.6f val %.6f" % (tl, vl))
    return m


if __name__ == "__main__":
    main()
