"""WaveGlow training step (SpeechSynthesis/Tacotron2, `-m WaveGlow`) on the gfx950 library: SURVEY.md 8 row f1."""
