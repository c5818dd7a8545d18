"""The two registered tasks (reference gym_minigrid/envs/twoarmy_v4.py, twoarmy_v6.py).  All dynamics
live in the HIP engine (csrc/twoarmy_engine.hip); these classes only pick the variant."""
from ..minigrid import MiniGridEnv


class Twoarmy_v6(MiniGridEnv):
    """Easy env: row-8 balls, fixed wall drop (twoarmy_v6.py)."""
    variant = 6


class Twoarmy_v4(MiniGridEnv):
    """Hard env: + RNG wall offsets and two RNG-gated patrol groups in room 2 (twoarmy_v4.py)."""
    variant = 4
