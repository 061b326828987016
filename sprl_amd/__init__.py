"""sprl_amd — MI355X-native self-play data-generation engine (drop-in for the sprl C++ workers)."""
