#!/usr/bin/env python
"""`python train_chaos.py --session=Experiment --tag=... --do_loss_ent --do_decoder_consistency --do_aux_path --do_memory`
-- same command line as the reference's train_chaos.py (README.md:63), running on MI355X via pacingpseudo_amd."""
from pacingpseudo_amd.train import train_main

if __name__ == '__main__':
    train_main()
