"""CPU oracle of the Max-Sum hot path -- test infrastructure only."""
