"""asvd4llm_amd — MI355X-native hot path of ASVD4LLM (activation-aware SVD compression) behind the reference's Python API."""
__version__ = "0.1.0"
