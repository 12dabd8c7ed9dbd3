"""CPU oracle of the flame hot path — TEST INFRASTRUCTURE ONLY (see oracle/flame_ref.h)."""
