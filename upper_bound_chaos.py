"""Launcher with the reference's script name: `python upper_bound_chaos.py --tag ... ` (fully-supervised upper bound).
The implementation lives in pacingpseudo_amd/upper_bound.py."""
from pacingpseudo_amd.upper_bound import train_main

if __name__ == '__main__':
    train_main()
