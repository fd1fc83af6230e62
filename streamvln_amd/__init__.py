"""streamvln_amd: MI355X-native StreamVLN streaming-inference path (HIP engine + Python host mirror)."""
from .config import StreamVLNConfig, TRUE, TINY, TRUE1, CONFIGS  # noqa: F401
