"""Entry point of the PPO + predictor variant (reference soa/train_ppo_predictor.py): same loop as
train_ppo with the `ppo_predictor` agent (frozen encoder/LSTM/decoder, 8-frame actor/critic)."""
from .train_ppo import main as _main


def main(argv=None):
    return _main(argv, predictor=True)


if __name__ == "__main__":
    main()
