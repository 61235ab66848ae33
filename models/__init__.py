"""Import-path shim of the drop-in boundary: the reference's trainers and evaluation scripts do
`from models.cavp_model import CAVP` / `SoundBank` (main_vpo_mono.py:98, trainer/trainer_cavp_vpo_mono.py:29).  Putting
this repository's root on PYTHONPATH ahead of the reference's makes those imports resolve to the MI355X implementation."""
