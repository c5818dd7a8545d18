"""Entry point of the self-orientation agent (reference soa/train_SoA.py): the train_ppo loop with the
`self_orinetation_agent` (frozen world model, 8-frame actor / critic with the predicted 3-step displacement in the
goal, orientation head trained on goal-reaching trajectories) and the vectorised SoA trainer."""
from .train_ppo import main as _main


def main(argv=None):
    return _main(argv, soa=True)


if __name__ == "__main__":
    main()
