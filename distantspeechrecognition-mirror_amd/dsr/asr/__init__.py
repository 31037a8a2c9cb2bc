"""Python face of the Millennium ASR classes on the hot path (asr/*/*.i names)."""
