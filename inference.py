"""Launcher with the reference's script name: `python inference.py --fold 0 --checkpoint_file <run dir or .pth> ...`
(Dice + HD95 of a checkpoint on the test fold).  The implementation lives in pacingpseudo_amd/inference.py."""
from pacingpseudo_amd.inference import main

if __name__ == '__main__':
    main()
