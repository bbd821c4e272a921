"""Only the config helpers the inference drivers import (reference train/parse_cfg.py); training is out of scope."""
