"""h5py is imported (run_pretraining.py:30) but unused on the lddl path."""
