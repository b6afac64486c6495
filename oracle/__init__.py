"""Test-infrastructure package: CPU oracle for the Sella hot path (see oracle/README.md)."""
