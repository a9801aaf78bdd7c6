"""Tacotron2 training step (SpeechSynthesis/Tacotron2, `-m Tacotron2`) on the gfx950 library: SURVEY.md 8 row f1, second half."""
