"""CPU oracle of the reference's hot path — test infrastructure, never imported by easykv_amd/."""
