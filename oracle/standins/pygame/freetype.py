"""Empty stand-in for pygame.freetype."""
