"""mo_yolo_amd: MI355X-native per-frame tracking inference path (DecoderTracker hot path)."""
__version__ = "0.1.0"
