"""The reference's train/DMsH_LN/loss.py:10-72 is a copy of DSPH's HyP module that its trainer imports but never calls
(train/DMsH_LN/hash_train.py:8); the name resolves to the one built for DSPH."""
from train.DSPH.loss import HyP  # noqa: F401
