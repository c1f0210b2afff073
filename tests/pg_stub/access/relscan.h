#include "postgres.h"
