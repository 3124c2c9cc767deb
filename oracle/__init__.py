"""CPU oracle of the smelter-render rasteriser passes — test infrastructure only."""
